"""Full resume state of a training run.

The reference saves only ``G_ema.state_dict()`` every ``save`` iterations (implementations/StyleGAN2/utils.py:119-123,
implementations/StyleGAN3/utils.py:83-85); that file format is kept (``save_generator`` / any ``G_*.pt`` of the reference loads
into this package's ``Generator``: identical keys).  A run at MI355X speed also needs to stop and continue, so ``save`` /
``load`` carry everything an iteration depends on: G, G_ema, D, both Adam states, the iteration counter, the path-length mean,
the ADA probability / statistic and the random-generator states of the rank."""
import torch


def save_generator(G_ema, path):
    """The reference's checkpoint: the EMA generator's state_dict."""
    torch.save(G_ema.state_dict(), path)


def _ada_of(step):
    ada = getattr(step, 'ada', None)
    if ada is None and hasattr(getattr(step, 'augment', None), 'update_p'):
        ada = step.augment
    return ada


def state(step):
    """Resume state of a ``TrainStep`` (StyleGAN2 or StyleGAN3 / ADA flavour) as a plain dict of tensors and numbers."""
    out = dict(G=step.G.state_dict(), G_ema=step.G_ema.state_dict() if step.G_ema is not None else None, D=step.D.state_dict(),
               optimizer_G=step.optimizer_G.state_dict(), optimizer_D=step.optimizer_D.state_dict(),
               batches_done=step.batches_done, pl_mean=getattr(step, 'pl_mean', 0.),
               rng_cpu=torch.get_rng_state(),
               rng_cuda=torch.cuda.get_rng_state() if torch.cuda.is_available() else None)
    ada = _ada_of(step)
    if ada is not None:
        out['ada'] = dict(state=ada.state_dict(), num_iter=ada._num_iter)
    return out


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def save(step, path):
    """Rank 0 writes the replicated state (networks, optimizers, counters, ADA); under data parallelism every rank also writes its
    own random-generator state next to it (``path + '.rng<rank>'``), then all ranks meet at a barrier -- no two processes ever write
    the same file."""
    rank, world = _rank_world()
    st = state(step)
    st['world_size'] = world
    if world > 1 and 'ada' in st and 'signsum' in st['ada']['state']:
        # the ADA sign statistic is accumulated per rank between two p updates and all-reduced at the update (nnutils/ada.py): store the
        # mean over the ranks, so that every rank resuming from rank 0's file carries 1 / world of the total (every rank calls save())
        import torch.distributed as dist
        tot = st['ada']['state']['signsum'].detach().clone()
        dist.all_reduce(tot)
        st['ada']['state'] = dict(st['ada']['state'], signsum=tot / world)
    if world > 1:
        # tagged with the iteration and the world size: load() refuses a side file of another save (a crash between the two writes would
        # otherwise pair a stale generator state with a new checkpoint)
        torch.save(dict(rng_cpu=st['rng_cpu'], rng_cuda=st['rng_cuda'], batches_done=st['batches_done'], world_size=world, rank=rank),
                   f'{path}.rng{rank}')
    if rank == 0:
        torch.save(st, path)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def load(step, path_or_state, map_location=None):
    """Restore a ``TrainStep`` in place; the next call of ``step(real)`` continues the run it was saved from."""
    st = torch.load(path_or_state, map_location=map_location, weights_only=False) if isinstance(path_or_state, str) else path_or_state
    step.G.load_state_dict(st['G'])
    step.D.load_state_dict(st['D'])
    if step.G_ema is not None and st.get('G_ema') is not None:
        step.G_ema.load_state_dict(st['G_ema'])
    step.optimizer_G.load_state_dict(st['optimizer_G'])
    step.optimizer_D.load_state_dict(st['optimizer_D'])
    step.batches_done = int(st['batches_done'])
    if hasattr(step, 'pl_mean'):
        step.pl_mean = st.get('pl_mean', 0.)
    ada = _ada_of(step)
    if 'ada' in st and 'signsum' in st['ada']['state']:
        # the stored sign statistic is one rank's share (total / saved world size): a run resumed on another number of ranks carries
        # total / current world size per rank, so that the all-reduced statistic of the first interval is the saved total
        saved_world, cur_world = int(st.get('world_size', 1)), _rank_world()[1]
        if saved_world != cur_world:
            sd = dict(st['ada']['state'])
            sd['signsum'] = sd['signsum'] * (saved_world / cur_world)
            st = dict(st, ada=dict(st['ada'], state=sd))
    if 'ada' in st:
        if ada is not None:
            ada.load_state_dict(st['ada']['state'])
            ada._num_iter = int(st['ada']['num_iter'])
        elif getattr(step, 'policy', None) == 'ada':
            # the StyleGAN2 trainer builds its pipe on the first batch (whose size fixes the p step): keep the state for that moment
            step._pending_ada_state = st['ada']
        else:
            raise RuntimeError('the checkpoint carries ADA state (p, sign statistic) but this trainer has no ADA pipe to restore it into')
    rank, world = _rank_world()
    if world > 1:
        import os
        import warnings
        side = f'{path_or_state}.rng{rank}' if isinstance(path_or_state, str) else None
        own = None
        if side is not None and os.path.exists(side):
            own = torch.load(side, map_location='cpu', weights_only=False)
            legacy = 'world_size' not in st and 'batches_done' not in own       # files of the format before the tags existed
            if not legacy and (int(own.get('batches_done', -1)) != int(st['batches_done']) or int(own.get('world_size', -1)) != world):
                own = None                                   # written by another save, or for another world size
        if own is not None:
            st = dict(st, rng_cpu=own['rng_cpu'], rng_cuda=own['rng_cuda'])
        elif rank != 0 or int(st.get('world_size', world)) != world:
            # no generator state of THIS rank for THIS checkpoint: restoring rank 0's state on every rank would give all replicas the same
            # noise and augmentation draws.  Re-seed deterministically per rank instead, and say so.
            warnings.warn(f'checkpoint: no matching random-generator state for rank {rank} (world size {world}); re-seeding this rank')
            torch.manual_seed(1234 + 1000003 * int(st['batches_done']) + rank)
            st = dict(st, rng_cpu=None, rng_cuda=None)
    if st.get('rng_cpu') is not None:
        torch.set_rng_state(st['rng_cpu'].cpu())
    if st.get('rng_cuda') is not None and torch.cuda.is_available():
        torch.cuda.set_rng_state(st['rng_cuda'].cpu())
    return step
