"""torch.profiler summary of one SG2 training step: which ATen ops (the glue around the HIP kernels) cost what."""
import sys, os, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from animeface_amd.implementations.StyleGAN2 import utils as U, model as M
from animeface_amd.nnutils import sample_nnoise, update_ema
dev = torch.device('cuda')
torch.manual_seed(0)
G, G_ema, D = M.Generator(256).to(dev), M.Generator(256).to(dev), M.Discriminator(256).to(dev)
G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); update_ema(G, G_ema, decay=0)
oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8)
step = U.TrainStep(G, G_ema, D, oG, oD, 10., 0., 16, 8, 'color,translation', 512, functools.partial(sample_nnoise, device=dev))
real = torch.rand(64, 3, 256, 256, device=dev) * 2 - 1
for _ in range(3): step(real)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(2): step(real)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=60))
