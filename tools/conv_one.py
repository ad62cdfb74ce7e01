import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2.conv import conv2d_fwd_raw, conv2d_wgrad_raw
N, Cin, Cout, H, W = [int(v) for v in sys.argv[1:6]]
mode = sys.argv[6] if len(sys.argv) > 6 else 'fwd'
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = 'cuda'
x = torch.randn(N, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5).to(torch.bfloat16)
dy = torch.randn(N, Cout, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
for _ in range(reps):
    if mode == 'fwd':
        conv2d_fwd_raw(x, w)
    else:
        conv2d_wgrad_raw(x, dy, 3)
torch.cuda.synchronize()
