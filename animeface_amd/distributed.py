"""Data-parallel gradient exchange for the StyleGAN2 loop: one process per GPU, RCCL over xGMI
(``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests).

The reference is single-GPU, so this is new design, shaped by the loop's needs:
  * G and D are reduced independently (the D-step only produces D gradients, the G-step only G gradients
    because D is frozen there), each with its own ``GradReducer``;
  * parameters that never receive a gradient (``InjectNoise.scale``, reference F10) and the
    ``autograd.grad(create_graph=True)`` pass of R1 must not confuse the reducer: buckets are launched from
    post-accumulate-grad hooks when complete, and ``finish()`` launches whatever is still pending;
  * gradients live in flat fp32 bucket buffers (``param.grad`` are views), so a bucket is all-reduced in place
    with no packing copies; buckets are filled in reverse registration order = backward order, so the
    all-reduce of early buckets overlaps the rest of backward (RCCL runs on its own stream; the compute
    stream only waits in ``finish()``, right before ``optimizer.step()``);
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by one link, so buckets
    are large (default 32 MiB -> 3 buckets for D's 85.6 MB, 3 for G's 77.4 MB) to amortise latency while
    still leaving two thirds of the exchange overlappable.
"""
import os

import torch
import torch.distributed as dist


def init_distributed():
    """Initialise from the torchrun environment; returns (rank, world_size, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('AGF_SINGLE_DEVICE') == '1':
        local_rank = 0                      # test hook: several ranks share GPU 0 (needs the gloo backend)
    force = os.environ.get('AGF_FORCE_DP') == '1'   # test hook: a ONE-rank process group, so that the RCCL path (process group, watchdog
    #                                                  thread, exchange between graph launches) can be exercised on a single-GPU box
    if (world > 1 or force) and not dist.is_initialized():
        backend = os.environ.get('AGF_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if force:
            os.environ.setdefault('MASTER_PORT', '29531')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class GradReducer:
    """Accumulation contract: between ``zero_grad()`` and ``finish()`` every parameter may receive ONE gradient while the reducer is
    enabled (a second one raises: its bucket's all-reduce may already be in flight).  To accumulate several backward passes set
    ``reducer.enabled = False`` for all but the last one: the hooks then only note which parameters were touched, autograd accumulates
    into ``param.grad`` as usual, the last (enabled) pass launches the buckets it completes and ``finish()`` packs and launches the
    rest -- including parameters that only the earlier passes touched."""

    def __init__(self, params, bucket_bytes=32 << 20, group=None, never_used=(), tail_bytes=8 << 20):
        """``never_used``: parameters the forward pass is known never to touch (``InjectNoise.scale``, reference F10).  A bucket is
        launched from the backward hooks once ALL of its parameters have their gradient, so such parameters would hold every bucket
        they sit in back until ``finish()`` (no overlap at all for the generator); they get a bucket of their own instead."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collectives = self.world > 1 or (dist.is_initialized() and os.environ.get('AGF_FORCE_DP') == '1')
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []          # each: dict(flat, params, pending, work)
        self._hooks = []
        self._touched = set()
        self.stats = dict(steps=0, buckets_from_hooks=0, buckets_at_finish=0, exposed_ms=0.0)
        self.measure = False             # bench.py: time the compute stream's wait in finish() with events
        self.recording = False           # set by GraphedTrainStep while the iteration is recorded (dp_mode 'ingraph')
        self.early = True                # launch complete buckets from the backward hooks (False: all at finish(); the graph-segmented loop)
        skip = {id(p) for p in never_used}
        idle = [p for p in self.params if id(p) in skip]
        if idle:
            self._make_bucket(idle)
        # backward produces gradients roughly in reverse parameter order.  The bucket that completes LAST cannot overlap anything -- its
        # all-reduce is the exposed tail of the exchange -- so it is kept small (``tail_bytes``); the buckets before it are large
        # (xGMI ring: per-link bound, latency amortised over 32 MiB).  Built from the tail end: walk the parameters in FORWARD order.
        order = [p for p in self.params if id(p) not in skip]
        groups, cur, cur_bytes, cap = [], [], 0, min(tail_bytes, bucket_bytes)
        for p in order:
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > cap:
                groups.append(cur)
                cur, cur_bytes, cap = [], 0, bucket_bytes
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for grp in reversed(groups):                     # bucket order = completion order in backward
            self._make_bucket(list(reversed(grp)))
        self.enabled = True

    @property
    def capturable(self):
        """True when the all-reduce itself can be recorded into a HIP graph: RCCL collectives are stream-ordered kernels on RCCL's own
        stream (torch forks it off the capturing stream and ``work.wait()`` joins it), so a captured iteration keeps the hook-launched,
        backward-overlapped exchange.  gloo runs on host threads and cannot be captured."""
        return (not self.collectives) or dist.get_backend(self.group) == 'nccl'

    def _make_bucket(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=torch.float32, device=plist[0].device)
        b = dict(flat=flat, params=plist, views=[], pending=len(plist), work=None, launched=False, early=False, packed=False)
        off = 0
        for p in plist:
            assert p.dtype == torch.float32
            view = flat[off:off + p.numel()].view_as(p)
            b['views'].append(view)
            p.grad = view
            off += p.numel()
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(b)))
        self.buckets.append(b)

    def _make_hook(self, bucket):
        def hook(param):
            # (recorded while disabled too: a parameter that only the earlier passes of an accumulation touch still carries a gradient,
            #  which finish() must neither drop nor leave out of the exchange)
            self._touched.add(id(param))
            if not self.enabled:
                return
            bucket['pending'] -= 1
            # one backward per parameter between zero_grad() and finish(): a second one would launch the bucket with a partial gradient
            # and then accumulate into a buffer whose all-reduce is in flight
            if bucket['pending'] < 0 or bucket['launched']:
                raise RuntimeError('GradReducer: a parameter received a second gradient before finish() / zero_grad() '
                                   '(accumulating several backward passes needs reducer.enabled = False until the last one)')
            if bucket['pending'] == 0 and self.early:
                bucket['early'] = True
                self._pack(bucket)
                self._launch(bucket)
        return hook

    def _launch(self, bucket):
        if bucket['launched']:
            return
        bucket['launched'] = True
        if self.collectives:
            if self.recording and not torch.cuda.is_current_stream_capturing():
                # a collective issued from a stream outside the capture while RCCL's stream is inside it would be recorded into the graph
                # AND handed to the process group's watchdog, which aborts the process when it polls the recorded event: fail here instead
                raise RuntimeError('GradReducer: a bucket became complete on a stream that is not being recorded '
                                   f'(stream {torch.cuda.current_stream().cuda_stream:#x}) while the iteration is recorded into a HIP graph')
            if dist.get_backend(self.group) == 'nccl':
                # RCCL averages inside the reduction: no extra pass over the bucket
                bucket['work'] = dist.all_reduce(bucket['flat'], op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            else:
                bucket['flat'].div_(self.world)
                bucket['work'] = dist.all_reduce(bucket['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def drain(self):
        """Before an iteration is recorded into a HIP graph: wait until the process group's watchdog thread has retired every collective
        issued so far.  The watchdog polls (every 100 ms) the end events of the collectives it still holds; such a poll, landing after
        RCCL's stream has joined the capture, aborts the process with hipErrorCapturedEvent -- the first recorded all-reduce comes within
        ~50 ms of the start of the recording, so without this the capture lost that race in about one run in ten."""
        if self.collectives and dist.get_backend(self.group) == 'nccl':
            torch.cuda.synchronize()
            pg = self.group if self.group is not None else dist.distributed_c10d._get_default_group()
            pg._wait_for_pending_works()

    def zero_grad(self):
        """``grad = None`` for every parameter: the backward pass then hands each gradient tensor over as it is (no accumulation launch
        per parameter into the bucket -- ~500 small adds per iteration for the two StyleGAN2 networks); ``_pack`` moves the gradients
        of a complete bucket into its flat buffer with ONE multi-tensor copy and re-points ``param.grad`` at the bucket views.  The slice
        of a parameter that receives no gradient keeps stale (finite) values; it is all-reduced and never read (``grad`` stays None)."""
        for b in self.buckets:
            b['pending'] = len(b['params'])
            b['work'] = None
            b['launched'] = False
            b['early'] = False
            b['packed'] = False
            for p in b['params']:
                p.grad = None
        self._touched.clear()

    def _pack(self, bucket):
        if bucket['packed']:
            return
        bucket['packed'] = True
        src, dst = [], []
        for p, v in zip(bucket['params'], bucket['views']):
            g = p.grad
            if g is None or g is v:
                continue
            src.append(g.detach())
            dst.append(v)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)

    def pack_all(self):
        """Graph-replayed training: called at the end of the captured backward segment, so the copies into the buckets are part of the
        graph and the optimizer step recorded later reads the bucket views."""
        for b in self.buckets:
            self._pack(b)

    def finish(self):
        """Launch incomplete buckets (parameters without gradient this step), wait for all of them, and give the parameters that
        received NO gradient ``grad = None`` -- as in the single-process loop (``zero_grad(set_to_none=True)``) Adam then skips them
        instead of stepping them with g = 0 (their step count and second-moment decay would otherwise differ from the reference:
        ``InjectNoise.scale`` always, D's last bias on lazy-R1 iterations)."""
        ev0 = ev1 = None
        if self.measure and self.buckets and self.buckets[0]['flat'].is_cuda and not torch.cuda.is_current_stream_capturing():
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for b in self.buckets:
            self.stats['buckets_from_hooks' if b['early'] else 'buckets_at_finish'] += 1
            if not b['launched']:
                self._pack(b)
                self._launch(b)
        for b in self.buckets:
            if b['work'] is not None:
                b['work'].wait()
                b['work'] = None
        if ev0 is not None:
            ev1.record()
            self._pending_events = getattr(self, '_pending_events', []) + [(ev0, ev1)]
        self.stats['steps'] += 1
        if self.enabled:
            for b in self.buckets:
                for p in b['params']:
                    if id(p) not in self._touched:
                        p.grad = None

    def launch_all(self):
        """Graph-replayed training (``GraphedTrainStep``, segmented mode): the backward pass ran inside a HIP graph, where no hook fires, so
        every bucket's all-reduce is issued here, right after that graph's launch: on the collective's own stream, ordered behind the
        graph by the event the process group records on the current stream.  What the caller launches next on the compute stream runs
        beside the exchange; ``wait_all()`` makes the compute stream wait for it.  Which parameters received a gradient is a static
        property of the captured iteration kind (``detach_untouched`` ran when it was captured)."""
        for b in self.buckets:
            b['launched'] = False
            self._launch(b)
        self.stats['steps'] += 1
        self.stats['buckets_at_finish'] += len(self.buckets)

    def wait_all(self):
        """The compute stream waits for the all-reduces issued by ``launch_all()``.  With ``measure`` the wait is bracketed by two events
        on the compute stream: their distance is the time the compute stream had nothing to run because of the exchange (the exposed
        part; 0 when the work launched in between outlasts the collective)."""
        ev0 = ev1 = None
        if self.measure and self.buckets and self.buckets[0]['flat'].is_cuda:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for b in self.buckets:
            if b['work'] is not None:
                b['work'].wait()
                b['work'] = None
        if ev0 is not None:
            ev1.record()
            self._pending_events = getattr(self, '_pending_events', []) + [(ev0, ev1)]

    def exchange_all(self):
        """``launch_all()`` + ``wait_all()`` with nothing in between: the whole exchange is exposed."""
        self.launch_all()
        self.wait_all()

    def detach_untouched(self):
        """``grad = None`` for the parameters whose hook did not fire since ``zero_grad()`` (second half of ``finish()``)."""
        if self.enabled:
            for b in self.buckets:
                for p in b['params']:
                    if id(p) not in self._touched:
                        p.grad = None

    def reset_stats(self):
        """Forget the counters and pending timing events (bench.py: right before its timed window, so that the report covers the window
        only -- not the eager warm-up iterations or the first, node-uploading replay of each recorded graph)."""
        self._pending_events = []
        self.stats = dict(steps=0, buckets_from_hooks=0, buckets_at_finish=0, exposed_ms=0.0)

    def overlap_report(self):
        """Counters for bench.py: how many buckets were launched from backward hooks (overlappable) vs. only at ``finish()``, and the time
        the compute stream spent waiting for the exchange in ``finish()`` (the exposed, non-overlapped part)."""
        ev = getattr(self, '_pending_events', [])
        if ev:
            torch.cuda.synchronize()
            self.stats['exposed_ms'] += sum(a.elapsed_time(b) for a, b in ev)
            self._pending_events = []
        s = dict(self.stats)
        s['buckets'] = len(self.buckets)
        s['bucket_mib'] = [round(b['flat'].numel() * 4 / 2 ** 20, 1) for b in self.buckets]
        s['exposed_ms_per_step'] = round(s['exposed_ms'] / max(s['steps'], 1), 4) if s['exposed_ms'] > 0 else None
        return s

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def never_used_parameters(module):
    """Parameters that exist only for state_dict parity and never receive a gradient (``InjectNoise.scale``, reference
    implementations/StyleGAN2/model.py:81-88): pass them to ``GradReducer(never_used=...)``."""
    out = []
    for m in module.modules():
        if type(m).__name__ == 'InjectNoise' and hasattr(m, 'scale'):
            out.append(m.scale)
    return out


def broadcast_module(module, src=0, group=None):
    """Start every replica from rank ``src``'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def check_replica_consistency(module, group=None, rtol=0.0, atol=0.0):
    """Replica-consistency assert in the spirit of the reference's vestigial ``check_ddp_consistency``
    (thirdparty/stylegan3_ops/misc.py:175-186): broadcast rank 0's tensor and compare."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return True
    for name, t in list(module.named_parameters()) + list(module.named_buffers()):
        other = t.detach().clone()
        dist.broadcast(other, src=0, group=group)
        if not torch.allclose(t.detach(), other, rtol=rtol, atol=atol, equal_nan=True):
            raise AssertionError(f'replica mismatch in {name}')
    return True
