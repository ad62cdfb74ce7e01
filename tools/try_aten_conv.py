import torch, time, itertools
import torch.nn.functional as F
dev='cuda'
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
B=32
for (C,Co,H,k) in [(64,64,259,3),(64,64,257,1),(256,256,67,3),(512,512,19,3)]:
    for xcl, wcl in itertools.product([False,True],[False,True]):
        x = torch.randn(B,C,H,H,device=dev,dtype=torch.bfloat16)
        w = torch.randn(Co,C,k,k,device=dev,dtype=torch.bfloat16)
        if xcl: x = x.contiguous(memory_format=torch.channels_last)
        if wcl: w = w.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True); w.requires_grad_(True)
        def f():
            y = F.conv2d(x,w,stride=2)
            return y
        tf = bench(f)
        y = f(); dy = torch.randn_like(y)
        def b():
            torch.autograd.grad(y,[x,w],dy,retain_graph=True)
        tb = bench(b)
        print(f'C{C} Co{Co} H{H} k{k} x_cl={xcl} w_cl={wcl}: fwd {tf:.2f} ms  bwd {tb:.2f} ms  y_cl={y.is_contiguous(memory_format=torch.channels_last)}', flush=True)
