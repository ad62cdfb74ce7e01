"""Probe: which stream / capture status does a post-accumulate-grad hook see while a backward pass is recorded into a HIP graph, and which
way of issuing an RCCL all_reduce from that hook survives capture + replay with the process group's watchdog thread running.
    python tools/probe/rccl_capture_probe.py A|B|C      (one-rank RCCL group)
A: dist.all_reduce(async_op=True) from the hook (torch forks its own RCCL stream)
B: own comm stream forked with wait_stream, dist.all_reduce(async_op=False) with the comm stream current, joined before the optimizer
C: as B but the hook re-enters the capturing stream explicitly first"""
import os
import sys
import threading
import time
import torch
import torch.distributed as dist

variant = sys.argv[1]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(256, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(), torch.nn.Linear(256, 1)).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True)
x = torch.randn(64, 256, device=dev)
comm = torch.cuda.Stream()
works, info = [], []
state = dict(capture_stream=None, forked=False)


def hook(p):
    cur = torch.cuda.current_stream()
    info.append((threading.get_ident(), cur.cuda_stream, torch.cuda.is_current_stream_capturing()))
    g = p.grad
    if variant == 'A':
        works.append(dist.all_reduce(g, op=dist.ReduceOp.AVG, async_op=True))
    else:
        if variant == 'C' and state['capture_stream'] is not None:
            cur = state['capture_stream']
        comm.wait_stream(cur)
        with torch.cuda.stream(comm):
            dist.all_reduce(g, op=dist.ReduceOp.AVG, async_op=False)
        state['forked'] = True


for p in net.parameters():
    p.register_post_accumulate_grad_hook(hook)


def body():
    opt.zero_grad(set_to_none=True)
    loss = net(x).square().mean()
    loss.backward()
    for w in works:
        w.wait()
    works.clear()
    if state['forked']:
        torch.cuda.current_stream().wait_stream(comm)
        state['forked'] = False
    opt.step()
    return loss.detach()


print('main thread', threading.get_ident(), 'default stream', torch.cuda.current_stream().cuda_stream, flush=True)
for _ in range(2):
    body()
torch.cuda.synchronize()
print('eager hooks:', sorted(set(info)), flush=True)
info.clear()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    state['capture_stream'] = torch.cuda.current_stream()
    print('capture stream', state['capture_stream'].cuda_stream, flush=True)
    out = body()
state['capture_stream'] = None
print('captured hooks:', sorted(set(info)), flush=True)
time.sleep(2.0)                       # give the watchdog time to poll
w0 = [p.detach().clone() for p in net.parameters()]
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
time.sleep(1.0)
moved = max(float((a - b.detach()).abs().max()) for a, b in zip(w0, net.parameters()))
print(f'variant {variant}: replay ok, loss {float(out):.5f}, weights moved {moved:.3g}', flush=True)
dist.barrier()
dist.destroy_process_group()
print('done', flush=True)
