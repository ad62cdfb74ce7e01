"""World-size-2 gloo tests of the data-parallel gradient path (runs on CPU, no GPU needed)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from animeface_amd import distributed as dp
    r, w, _ = dp.init_distributed()
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                 # deliberately different init per rank
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    unused = torch.nn.Parameter(torch.zeros(1))   # like InjectNoise.scale: registered but never receives a gradient
    params = list(net.parameters()) + [unused]
    dp.broadcast_module(net)
    dp.check_replica_consistency(net)
    red = dp.GradReducer(params, bucket_bytes=64)  # tiny buckets -> several buckets, launched from hooks
    assert len(red.buckets) >= 2
    opt = torch.optim.SGD(params, lr=0.1)
    torch.manual_seed(7)                           # same data stream on both ranks, sharded by rank below
    data = torch.randn(4, 8, 6)
    for it in range(3):
        red.zero_grad()
        x = data[it, rank * 4:(rank + 1) * 4]      # per-rank shard of the global batch of 8
        # a double-backward term (like R1) must not trip the hooks
        xr = x.clone().requires_grad_(True)
        g, = torch.autograd.grad(net(xr).sum(), xr, create_graph=True)
        loss = net(x).square().mean() + 0.1 * g.square().mean()
        loss.backward()
        red.finish()
        opt.step()
    dp.check_replica_consistency(net)
    if rank == 0:
        torch.save({k: v.clone() for k, v in net.state_dict().items()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_reducer_matches_single_process_large_batch(tmp_path):
    out = str(tmp_path / 'dp.pt')
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method='spawn')
    dp_state = torch.load(out)
    # single-process reference: same init as rank 0, full batch of 8, loss = mean over the two shards' losses
    torch.manual_seed(100)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    torch.manual_seed(7)
    data = torch.randn(4, 8, 6)
    for it in range(3):
        opt.zero_grad()
        total = 0
        for r in range(2):
            x = data[it, r * 4:(r + 1) * 4]
            xr = x.clone().requires_grad_(True)
            g, = torch.autograd.grad(net(xr).sum(), xr, create_graph=True)
            total = total + (net(x).square().mean() + 0.1 * g.square().mean()) / 2
        total.backward()
        opt.step()
    for k, v in net.state_dict().items():
        torch.testing.assert_close(dp_state[k], v, rtol=1e-5, atol=1e-6)


def _ada_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from animeface_amd import distributed as dp
    from animeface_amd.nnutils.ada import ADA
    dp.init_distributed()
    torch.manual_seed(5)
    logits = torch.randn(8, 8, 1) + torch.linspace(1.5, -1.5, 8).reshape(8, 1, 1)     # global batch of 8 per iteration
    ada = ADA(4, 2, 1, 0.6)                                                            # per-rank batch 4
    traj = []
    for it in range(8):
        ada.update_p(logits[it, rank * 4:(rank + 1) * 4])
        traj.append(float(ada.p))
    if rank == 0:
        torch.save(traj, out)
    dist.barrier()
    dist.destroy_process_group()


def test_ada_probability_follows_the_global_batch(tmp_path):
    """Two ranks with half the batch each must walk the same p trajectory as one process with the whole batch
    (reference rule nnutils/ada.py:25-36 applied to the global batch)."""
    from animeface_amd.nnutils.ada import ADA
    out = str(tmp_path / 'ada.pt')
    mp.start_processes(_ada_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method='spawn')
    torch.manual_seed(5)
    logits = torch.randn(8, 8, 1) + torch.linspace(1.5, -1.5, 8).reshape(8, 1, 1)
    ada = ADA(8, 2, 1, 0.6)
    ref = []
    for it in range(8):
        ada.update_p(logits[it])
        ref.append(float(ada.p))
    assert torch.load(out) == pytest.approx(ref, abs=1e-7)
    assert max(ref) > 0
