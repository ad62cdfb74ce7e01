"""One StyleGAN3 filtered_lrelu layer (forward, or forward + backward) for profiling / timing.
    python tools/flr_one.py <layer 0..13> [fwd|bwd] [reps] [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN3 import model as M
from animeface_amd.stylegan3_ops import filtered_lrelu as FL
li = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = 'cuda'
G = M.Generator(256, 512)
l = G.synthesis.net[li].to(dev)
C = l.conv.weight.shape[0]
ch, sizes, *_ = M.get_layer_params(256, 14, 2 ** 14 * 0.5, 512, 3, 10)
S = int(sizes[max(li - 1, 0)]) + 2
x = torch.randn(B, C, S, S, device=dev).to(torch.bfloat16).requires_grad_(mode == 'bwd')
b = l.bias.detach().to(torch.bfloat16)
def run():
    y = FL.filtered_lrelu(x, l.up_filter, l.down_filter, b, l.up_factor, l.down_factor, l.padding, l.gain, l.negative_slope, l.conv_clamp)
    if mode == 'bwd':
        y.backward(torch.ones_like(y))
    return y
y = run(); torch.cuda.synchronize()
t = time.time()
for _ in range(reps): run()
torch.cuda.synchronize()
print(f'layer {li} {mode} x {tuple(x.shape)} -> y {tuple(y.shape)} up{l.up_factor} down{l.down_factor}: {(time.time()-t)/reps*1e3:.3f} ms')
