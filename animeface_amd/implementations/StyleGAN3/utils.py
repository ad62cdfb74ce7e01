"""StyleGAN3 training loop on MI355X.

Mirrors the reference's ``implementations/StyleGAN3/utils.py`` (``train`` :15-100, ``main`` :102-198): one ``G(z)`` per
iteration shared by the D-step (detached) and the G-step, R1 ADDED to the adversarial loss every ``gp_every`` iterations
(iteration 0 included, :50-53), the G-step through the already-updated D (:63-67), ``update_ema(copy_buffers=True)``
(:77), Adam with the mapping network at ``lr * map_lr_scale`` (:176-181), and the warm-up ``D(G(const_input))`` that
advances the ``ema`` / ``w_avg`` buffers once before training (:173-174).  Differences, none in a step's arithmetic:
  * bf16 activations instead of fp16 autocast + GradScaler (``amp`` selects bf16, otherwise fp32);
  * D's parameters are frozen during the G-step (the reference accumulates and then zeroes those gradients);
  * no per-iteration ``save_image`` / ``.item()`` host syncs (kept behind ``log_every`` / ``on_save``);
  * optional data parallelism through ``animeface_amd.distributed.GradReducer``.
"""
import functools

import torch
import torch.optim as optim

from ...nnutils import get_device, sample_nnoise, update_ema, freeze
from ...nnutils.loss import NonSaturatingLoss, r1_regularizer
from ...thirdparty.diffaugment import DiffAugment
from ... import distributed as dp
from .model import Generator, Discriminator
from ..StyleGAN2.conv import cached_weights, invalidate_cached, PrepPlan, recording_plans, ZeroArena, zero_arena


class TrainStep:
    """One iteration of the reference loop (utils.py:30-77)."""

    def __init__(self, G, G_ema, D, optimizer_G, optimizer_D, gp_lambda, gp_every, augment, latent_dim,
                 reducer_G=None, reducer_D=None):
        self.G, self.G_ema, self.D = G, G_ema, D
        self.optimizer_G, self.optimizer_D = optimizer_G, optimizer_D
        self.gp_lambda, self.gp_every, self.augment, self.latent_dim = gp_lambda, gp_every, augment, latent_dim
        self.reducer_G, self.reducer_D = reducer_G, reducer_D
        self.adv_fn = NonSaturatingLoss()
        self.gp_fn = r1_regularizer()
        self.batches_done = 0
        # prepared conv weights (bf16 OHWI copies in both orientations) live for one iteration and are made by one launch per network
        # from the second iteration on, as in StyleGAN2.utils.TrainStep (136 preparation launches per iteration without)
        self._plan_G, self._plan_D = PrepPlan(G.parameters()), PrepPlan(D.parameters())
        self._arena = ZeroArena()                      # zero-initialised fp32 scratch of the kernels that accumulate atomically: one fill per iteration

    def _mbsd_group_size(self):
        from .model import MinibatchStdDev
        for m in self.D.modules():
            if isinstance(m, MinibatchStdDev):
                return m.group_size
        return None

    @staticmethod
    def _zero(opt, reducer):
        if reducer is not None:
            reducer.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)

    def __call__(self, real):
        with cached_weights(), recording_plans(self._plan_G, self._plan_D), zero_arena(self._arena, real.device):
            self._plan_G.run()
            self._plan_D.run()
            out = self._iteration(real)
        self._plan_G.build(), self._plan_D.build()              # no-ops after the first iteration
        return out

    def _iteration(self, real):
        G, D = self.G, self.D
        self._zero(self.optimizer_D, self.reducer_D)
        self._zero(self.optimizer_G, self.reducer_G)
        z = sample_nnoise((real.size(0), self.latent_dim), real.device)

        fake = G(z)
        real_aug = self.augment(real)
        fake_aug = self.augment(fake)
        B, g = real.size(0), self._mbsd_group_size()
        if g is not None and B % g == 0:
            # one batch-2B pass; interleaving in chunks of B/g keeps the minibatch-stddev groups {m, m + B/g, ...} of the two
            # halves exactly as they are in separate passes (same argument as StyleGAN2.utils.TrainStep._d_half)
            both = torch.cat([c for pair in zip(real_aug.chunk(g), fake_aug.detach().chunk(g)) for c in pair])
            prob = D(both).reshape(g, 2, B // g, -1)
            real_prob, fake_prob = prob[:, 0].reshape(B, -1), prob[:, 1].reshape(B, -1)
        else:
            real_prob = D(real_aug)
            fake_prob = D(fake_aug.detach())
        D_loss = self.adv_fn.d_loss(real_prob, fake_prob)
        if self.gp_lambda > 0 and self.batches_done % self.gp_every == 0:
            D_loss = D_loss + self.gp_fn(real, D, None) * self.gp_lambda
        D_loss.backward()
        if self.reducer_D is not None:
            self.reducer_D.finish()
        self.optimizer_D.step()
        invalidate_cached(D.parameters())
        self._plan_D.run()

        for p in D.parameters():
            p.requires_grad_(False)
        G_loss = self.adv_fn.g_loss(D(fake_aug))
        G_loss.backward()
        for p in D.parameters():
            p.requires_grad_(True)
        if self.reducer_G is not None:
            self.reducer_G.finish()
        self.optimizer_G.step()

        update_ema(G, self.G_ema, copy_buffers=True)
        if hasattr(self.augment, 'update_p'):                 # ADA pipe (reference implementations/ADA/utils.py:73)
            self.augment.update_p(real_prob.detach())
        self.batches_done += 1
        return D_loss.detach(), G_loss.detach(), fake.detach()


class GraphedTrainStep:
    """The StyleGAN3 iteration replayed from HIP graphs: the body of ``TrainStep.__call__`` (G(z), both D passes, both backward passes, both
    fused Adam steps, EMA) is recorded once per iteration kind -- adversarial loss only, or with the R1 penalty of every ``gp_every``-th
    iteration (reference utils.py:50-53) -- and replayed with one host call.  Eagerly the ~2 700 launches of a 512x512 iteration take the
    host longer to issue than the GPU to run (84.7 ms of kernels in a 104.9 ms iteration, profiles/r04_final_sg3_512_kernel_stats.csv).
    Kernels, their order and their arithmetic are those of the eager step; random draws come from torch's graph-safe generator state.
    Needs capturable optimizers (``build_optimizers(..., capturable=True)``), batches of one fixed shape, no gradient reducers (the
    data-parallel exchange of this trainer is issued from the host) and an augmentation without a host-side schedule (``update_p``)."""

    def __init__(self, step, real, warmup=2):
        if not real.is_cuda:
            raise RuntimeError('graph capture needs a GPU batch')
        if step.reducer_G is not None or step.reducer_D is not None:
            raise RuntimeError('StyleGAN3 graph capture: gradient reducers are not supported (run the eager TrainStep)')
        if hasattr(step.augment, 'update_p'):
            raise RuntimeError('StyleGAN3 graph capture: the ADA p schedule reads D(real) on the host (run the eager TrainStep)')
        for opt in (step.optimizer_G, step.optimizer_D):
            if not all(g.get('capturable', False) for g in opt.param_groups):
                raise RuntimeError('graph capture needs capturable optimizers: build_optimizers(..., capturable=True)')
        self.step, self.graphs = step, {}
        self.static_real = real.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # eager iterations first: optimizer state, workspaces and caches reach their final size
            for _ in range(warmup):
                step(self.static_real)
        torch.cuda.current_stream().wait_stream(side)

    @property
    def batches_done(self):
        return self.step.batches_done

    def _kind(self, it):
        st = self.step
        return 'r1' if (st.gp_lambda > 0 and it % st.gp_every == 0) else 'gan'

    def kinds(self):
        return set(self.graphs)

    def _capture(self, it):
        kind = self._kind(it)
        if kind in self.graphs:
            return
        st = self.step
        saved = st.batches_done
        st.batches_done = it
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                out = st(self.static_real)                 # records; the Python body runs once and leaves batches_done advanced
        finally:
            st.batches_done = saved
        self.graphs[kind] = (graph, out)

    def capture_all(self):
        st = self.step
        for it in range(1, (st.gp_every if st.gp_lambda > 0 else 1) + 1):
            self._capture(it)

    def __call__(self, real):
        st = self.step
        it = st.batches_done
        self.static_real.copy_(real)
        self._capture(it)
        graph, out = self.graphs[self._kind(it)]
        graph.replay()
        st.batches_done = it + 1
        return out


GRAPH_AFTER = 2        # eager iterations at the start of train(graphs=True) before the iteration is recorded


def train(max_iters, dataset, latent_dim, const_input,
          G, G_ema, D, optimizer_G, optimizer_D,
          gp_lambda, gp_every, augment,
          device, amp, save=1000, log_file=None, log_every=50, on_save=None, reducer_G=None, reducer_D=None, resume=None,
          checkpoint_path=None, log=print, graphs=False):
    """Same positional signature as the reference's ``train`` (utils.py:15-20).  Every ``log_every`` iterations one line with the losses
    and the throughput since the previous line goes to ``log``.  ``graphs=True``: after ``GRAPH_AFTER`` ordinary eager iterations (optimizer
    state, workspaces and caches then have their final size) the iteration is recorded into HIP graphs and replayed (``GraphedTrainStep``;
    needs capturable optimizers); recording executes nothing, so no iteration is consumed."""
    import time
    from ... import distributed as _dp
    step = TrainStep(G, G_ema, D, optimizer_G, optimizer_D, gp_lambda, gp_every, augment, latent_dim, reducer_G, reducer_D)
    if resume is not None:                                  # full resume state (animeface_amd/checkpoint.py), not just G_ema
        from ... import checkpoint
        checkpoint.load(step, resume, map_location=device)
    runner, it_start = step, step.batches_done
    history = []
    t_last, it_last = time.perf_counter(), step.batches_done
    world = _dp.dist.get_world_size() if _dp.dist.is_initialized() else 1
    if log is not None:
        log(f'training on {torch.cuda.get_device_name(device) if torch.cuda.is_available() else device} | {"bf16" if amp else "fp32"} | world size {world}')
    while step.batches_done < max_iters:
        for real in dataset:
            real = real.to(device, non_blocking=True)
            it = step.batches_done
            if graphs and runner is step and real.is_cuda and step.batches_done - it_start >= GRAPH_AFTER:
                runner = GraphedTrainStep(step, real, warmup=0)
            D_loss, G_loss, fake = runner(real)
            if it % save == 0 and checkpoint_path is not None and it > 0:
                from ... import checkpoint
                checkpoint.save(step, checkpoint_path)
            if it % save == 0 and on_save is not None:
                with torch.no_grad():
                    on_save(it, G_ema(const_input), G_ema)
            if log_every and it % log_every == 0:
                d, g = D_loss.item(), G_loss.item()
                history.append((it, d, g))
                now = time.perf_counter()
                if log is not None and step.batches_done > it_last:
                    log(f'iter {it:7d} | D_loss {d:9.4f} | G_loss {g:9.4f} | {(step.batches_done - it_last) * real.size(0) * world / (now - t_last):8.1f} img/s')
                t_last, it_last = now, step.batches_done
            if step.batches_done == max_iters:
                break
    return history


def build_optimizers(G, D, lr, map_lr_scale, betas, capturable=False):
    """reference utils.py:176-181.  ``capturable``: step counters on the device, so that the steps can be recorded into a HIP graph."""
    fused = all(p.is_cuda for p in G.parameters())
    optimizer_G = optim.Adam([{'params': G.synthesis.parameters()},
                              {'params': G.map.parameters(), 'lr': lr * map_lr_scale}], lr=lr, betas=betas, fused=fused, capturable=capturable)
    optimizer_D = optim.Adam(D.parameters(), lr=lr, betas=betas, fused=fused, capturable=capturable)
    return optimizer_G, optimizer_D


def build_models(args, device, compute_dtype):
    mk_G = lambda: Generator(args.image_size, args.latent_dim, args.num_layers, args.map_num_layers, args.channels,
                             args.max_channels, args.style_dim, not args.no_pixel_norm, args.image_channels,
                             args.output_scale, args.margin_size, args.first_cutoff, args.first_stopband,
                             args.last_stopband_rel, args.kernel_size, compute_dtype=compute_dtype)
    G, G_ema = mk_G(), mk_G()
    freeze(G_ema)
    update_ema(G, G_ema, 0., copy_buffers=True)
    D = Discriminator(args.image_size, args.image_channels, args.d_channels, args.d_max_channels, 3, args.mbsd_group_size,
                      args.mbsd_channels, args.bottom, args.gaus_filter_size, 'lrelu', 1, compute_dtype=compute_dtype)
    return G.to(device), G_ema.to(device), D.to(device)


SG3_ARGS = dict(
    num_test=[16, 'number of images for eval'],
    image_channels=[3, 'number of image channels'],
    latent_dim=[512, 'latent dimension'],
    num_layers=[14, 'number of layers in G'],
    map_num_layers=[2, 'number of layers in mapping network'],
    channels=[32, 'channel base'],
    max_channels=[512, 'maximum channel width'],
    style_dim=[512, 'style code dimension'],
    kernel_size=[3, 'kernel size. 3'],
    no_pixel_norm=[False, 'no pixel normalization'],
    output_scale=[0.25, 'scale output tensor with'],
    margin_size=[10, 'bigger size to work on'],
    first_cutoff=[2., 'first cutoff'],
    first_stopband=[2 ** 2.1, 'first stopband'],
    last_stopband_rel=[2 ** 0.3, 'last relative stopband'],
    d_channels=[32, 'channel base for D'],
    d_max_channels=[512, 'maximum channels in D'],
    mbsd_group_size=[4, 'mini-batch stddev group size'],
    mbsd_channels=[1, 'mini-batch stddev channels'],
    bottom=[4, 'bottom width in D'],
    gaus_filter_size=[4, 'filter size in D'],
    lr=[0.0025, 'learning rate'],
    map_lr_scale=[0.01, 'scale learning rate for mapping network with'],
    betas=[[0., 0.99], 'betas'],
    gp_lambda=[3., 'lambda for r1'],
    gp_every=[16, 'calc penalty every'],
    policy=['color,translation', 'policy for DiffAugment'],
    log_every=[50, 'iterations between log lines (losses, img/s); not a flag of the reference'],
    hip_graphs=[False, 'replay the training iteration from HIP graphs (single GPU, DiffAugment)'],
    logfile=[str, 'log file'])


def main(parser, dataset=None):
    """``implementations.StyleGAN3.main(parser)`` contract of the reference's main.py.  The dataset is injected (an
    iterable of image batches in [-1, 1]); without one a synthetic uniform batch is cycled."""
    from ...utils_argument import add_args
    parser = add_args(parser, SG3_ARGS)
    args = parser.parse_args()
    rank, world, _ = dp.init_distributed()
    device = get_device(not args.disable_gpu)
    amp = not args.disable_amp and not args.disable_gpu
    compute_dtype = torch.bfloat16 if amp else torch.float32
    const_input = sample_nnoise((args.num_test, args.latent_dim), device)
    G, G_ema, D = build_models(args, device, compute_dtype)
    dp.broadcast_module(G), dp.broadcast_module(G_ema), dp.broadcast_module(D)
    D(G(const_input))                                             # the reference's warm-up call; it moves ema / w_avg
    graphs = bool(args.hip_graphs) and world == 1 and device.type == 'cuda'
    optimizer_G, optimizer_D = build_optimizers(G, D, args.lr, args.map_lr_scale, tuple(args.betas), capturable=graphs)
    reducer_G = dp.GradReducer(G.parameters()) if world > 1 else None
    reducer_D = dp.GradReducer(D.parameters()) if world > 1 else None
    if dataset is None:
        gen = torch.Generator(device='cpu').manual_seed(rank)
        batch = (torch.rand(args.batch_size, args.image_channels, args.image_size, args.image_size, generator=gen) * 2 - 1).to(device)
        dataset = [batch]
    if args.max_iters < 0:
        args.max_iters = len(dataset) * args.default_epochs
    augment = functools.partial(DiffAugment, policy=args.policy)
    return train(args.max_iters, dataset, args.latent_dim, const_input, G, G_ema, D, optimizer_G, optimizer_D,
                 args.gp_lambda, args.gp_every, augment, device, amp, args.save, args.logfile, log_every=args.log_every,
                 reducer_G=reducer_G, reducer_D=reducer_D, graphs=graphs)
