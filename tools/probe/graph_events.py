import torch
x = torch.randn(4096, 4096, device='cuda')
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x @ x
torch.cuda.current_stream().wait_stream(s)
evs = []
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True, external=True); e1 = torch.cuda.Event(enable_timing=True, external=True)
        e0.record(); y = x @ x; e1.record()
        evs.append((e0, e1))
for rep in range(2):
    g.replay()
    torch.cuda.synchronize()
    print([round(a.elapsed_time(b), 3) for a, b in evs])
