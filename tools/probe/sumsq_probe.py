import sys, os, torch
sys.path.insert(0, os.getcwd())
from animeface_amd.implementations.StyleGAN2.conv import scale_dot_raw
from animeface_amd.implementations.StyleGAN3.model import mean_square
def t_(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for shape in [(16,512,36,36),(16,512,84,84),(16,362,148,148),(16,161,276,276),(16,72,532,532),(16,32,532,532)]:
    x=torch.randn(shape,device='cuda').to(torch.bfloat16)
    ref=lambda: torch.linalg.vector_norm(x,2,dtype=torch.float32).square()
    n=x.numel(); 
    def mine():
        m = n // 8 * 8
        v = x.view(-1)[:m].view(1, m // 8, 1, 8).permute(0, 3, 1, 2)      # [1, 8, m/8, 1] channels-last view of the flat buffer
        _, ds = scale_dot_raw(v, v, ones, want_dx=False)
        return ds.sum()
    ones=torch.ones(1,8,device='cuda')
    a=ref().item(); b=mine().item(); c=mean_square(x).item()*n
    tk=t_(lambda: mean_square(x))
    print(shape, 'MB', n*2/1e6, 'aten us', round(t_(ref),1), 'scale_dot us', round(t_(mine),1), 'agf_sum_squares (+ the slot sum) us', round(tk,1), 'TB/s', round(n*2/tk/1e6,2), 'rel', abs(a-b)/a, abs(a-c)/a)
