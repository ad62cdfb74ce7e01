import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2.conv import conv2d_wgrad_raw
N, Cin, Cout, H, W = [int(v) for v in sys.argv[1:6]]
KS = int(sys.argv[6]) if len(sys.argv) > 6 else 3
x = torch.randn(N, Cin, H, W, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
dy = torch.randn(N, Cout, H, W, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
si = (torch.rand(N, Cin, device='cuda') + 0.5) if os.environ.get('SCALED') == '1' else None
so = (torch.rand(N, Cout, device='cuda') + 0.5) if os.environ.get('SCALED') == '1' else None
for _ in range(3):
    conv2d_wgrad_raw(x, dy, KS, in_scale=si, out_scale=so)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    conv2d_wgrad_raw(x, dy, KS, in_scale=si, out_scale=so)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print(json.dumps(dict(shape=[N, Cin, Cout, H, W], ms=round(ms, 4), TFLOPs=round(2.0 * N * H * W * Cin * Cout * KS * KS / ms / 1e9, 1))))
