"""Time one 3x3 forward / data-gradient conv launch: python tools/time_conv.py N Cin Cout H W [mask|maskbits|bias|biasbits|demod|noise]
(bias: + bias + lrelu, a discriminator conv; demod: out_scale + bias + noise + lrelu, a generator conv after POSTSCALE_X/PRESCALE_G)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd import _lib
if os.environ.get('AGF_PROBE_LIB'):          # a probe build (tools/probe/build_variant.sh)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libagf_ops_%s.so' % os.environ['AGF_PROBE_LIB'])
from animeface_amd.implementations.StyleGAN2.conv import conv2d_fwd_raw, prep_weights_raw
N, Cin, Cout, H, W = [int(v) for v in sys.argv[1:6]]
mode = sys.argv[6] if len(sys.argv) > 6 else ''
mask = mode in ('mask', 'masknosum', 'maskbits')
x = torch.randn(N, Cin, H, W, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(Cout, Cin, 3, 3, device='cuda') / (Cin * 9) ** 0.5
wq = prep_weights_raw(w, 1.0, torch.bfloat16)[0]
my = torch.randn(N, Cout, H, W, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if mask else None
ms_ = torch.zeros(256, Cout, device='cuda') if mode in ('mask', 'maskbits') else None
mb = torch.randint(-2**31, 2**31 - 1, (N, H, W, Cout // 32), dtype=torch.int32, device='cuda') if mode == 'maskbits' else None
bo = torch.empty((N, H, W, Cout // 32), dtype=torch.int32, device='cuda') if mode == 'biasbits' else None
from animeface_amd.implementations.StyleGAN2.conv import ACT_LRELU
kw = {}
if mode in ('bias', 'demod', 'biasbits'):
    kw = dict(bias=torch.randn(Cout, device='cuda'), act=ACT_LRELU, gain=2 ** 0.5)
if mode == 'demod':
    kw.update(out_scale=torch.rand(N, Cout, device='cuda') + 0.5, noise=torch.randn(N, 1, H, W, device='cuda'))
def run():
    return conv2d_fwd_raw(x, wq, prepared=True, mask_y=None if mb is not None else my, mask_bits=mb, bits_out=bo, mask_sum=ms_, **kw)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
gb = (N * H * W * (Cin + Cout * (2 if (mask and mb is None) else 1)) * 2 + (N * H * W * Cout // 8 if (mb is not None or bo is not None) else 0) + Cout * Cin * 18) / 1e9
print(json.dumps(dict(shape=[N, Cin, Cout, H, W], mode=mode, ms=round(ms, 4), TFLOPs=round(2.0 * N * H * W * Cin * Cout * 9 / ms / 1e9, 1), TBps=round(gb / ms, 2))))
