"""The MFMA conv kernels at the shapes bench.py times (BASELINE config StyleGAN2 256x256, batch 64 / merged 128), against a CPU fp32
convolution of the same bf16-rounded operands -- computed by the oracle's side of the house (torch CPU), not by a GPU library.  The
forward kernels are launched on the FULL batch (that is what selects the tiling: persistent streaming kernel, 8-wave direct-to-LDS,
weight-stationary, multi-image tiles); the CPU reference is evaluated on a sample of the images (a convolution treats images
independently).  The weight gradient sums over the batch, so its reference uses every image."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# (N, Cin, Cout, H, W): forward / data-gradient shapes of the 256x256 step
FWD_SHAPES = [(64, 512, 512, 32, 32), (128, 512, 512, 16, 16), (64, 256, 256, 64, 64), (64, 128, 128, 128, 128), (128, 64, 64, 256, 256),
              (64, 64, 32, 256, 256), (64, 32, 32, 256, 256), (128, 32, 64, 256, 256), (64, 512, 512, 4, 4), (128, 256, 512, 32, 32),
              (128, 64, 128, 128, 128)]


def _mk(N, Cin, Cout, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(torch.bfloat16)
    b = torch.randn(Cout, generator=g)
    return x, w, b


@pytest.mark.parametrize('shape', FWD_SHAPES, ids=[f'N{s[0]}_{s[1]}to{s[2]}_{s[3]}x{s[4]}' for s in FWD_SHAPES])
@pytest.mark.parametrize('modulated', [False, True])
def test_conv_forward_at_bench_shapes_vs_cpu(shape, modulated):
    from animeface_amd.implementations.StyleGAN2.conv import conv2d_fwd_raw, ACT_LRELU
    N, Cin, Cout, H, W = shape
    x, w, b = _mk(N, Cin, Cout, H, W, 1)
    g = torch.Generator().manual_seed(2)
    s_in = torch.rand(N, Cin, generator=g) + 0.5 if modulated else None
    s_out = torch.rand(N, Cout, generator=g) + 0.5 if modulated else None
    nz = torch.randn(N, 1, H, W, generator=g) if modulated else None
    y = conv2d_fwd_raw(x.to(DEV).contiguous(memory_format=torch.channels_last), w.to(DEV), in_scale=None if s_in is None else s_in.to(DEV),
                       out_scale=None if s_out is None else s_out.to(DEV), bias=b.to(DEV), noise=None if nz is None else nz.to(DEV),
                       act=ACT_LRELU, alpha=0.2)
    # no atomics on this path: a second launch must give the same bits (a race in the hand-counted vmcnt / barrier protocol of the
    # streaming kernels would show up here long before it shows up in a tolerance)
    y2 = conv2d_fwd_raw(x.to(DEV).contiguous(memory_format=torch.channels_last), w.to(DEV), in_scale=None if s_in is None else s_in.to(DEV),
                        out_scale=None if s_out is None else s_out.to(DEV), bias=b.to(DEV), noise=None if nz is None else nz.to(DEV),
                        act=ACT_LRELU, alpha=0.2)
    assert torch.equal(y, y2)
    torch.cuda.synchronize()
    sample = sorted({0, 1, N // 2, N - 1})
    xs = x[sample].float()
    if modulated:
        xs = xs * s_in[sample][:, :, None, None]
    ref = F.conv2d(xs, w.float(), padding=1)
    if modulated:
        ref = ref * s_out[sample][:, :, None, None] + nz[sample]
    ref = F.leaky_relu(ref + b[None, :, None, None], 0.2)
    got = y[sample].float().cpu()
    err = (got - ref).abs()
    scale = ref.abs().max()
    # bf16 output rounding (2^-9 relative) + rounding of the style-scaled operand on the modulated layers
    assert float(err.max() / scale) < (1.2e-2 if modulated else 5e-3), float(err.max() / scale)
    assert float(err.square().mean().sqrt() / ref.square().mean().sqrt()) < (5e-3 if modulated else 2.5e-3)
    assert torch.isfinite(y).all()


@pytest.mark.parametrize('shape', [(64, 512, 512, 32, 32), (64, 64, 64, 256, 256), (64, 64, 32, 256, 256), (64, 128, 256, 64, 64)],
                         ids=lambda s: f'N{s[0]}_{s[1]}to{s[2]}_{s[3]}x{s[4]}')
def test_conv_weight_gradient_at_bench_shapes_vs_cpu(shape):
    from animeface_amd.implementations.StyleGAN2.conv import conv2d_wgrad_raw
    N, Cin, Cout, H, W = shape
    x, _, _ = _mk(N, Cin, Cout, H, W, 3)
    g = torch.Generator().manual_seed(4)
    dy = torch.randn(N, Cout, H, W, generator=g).to(torch.bfloat16)
    dw = conv2d_wgrad_raw(x.to(DEV).contiguous(memory_format=torch.channels_last), dy.to(DEV).contiguous(memory_format=torch.channels_last), 3)
    assert torch.equal(dw, conv2d_wgrad_raw(x.to(DEV).contiguous(memory_format=torch.channels_last),
                                            dy.to(DEV).contiguous(memory_format=torch.channels_last), 3))       # two-stage combine: no atomics
    wz = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    ref = torch.zeros(Cout, Cin, 3, 3)
    for n0 in range(0, N, 8):                                            # in chunks: bounded host memory
        (gw,) = torch.autograd.grad(F.conv2d(x[n0:n0 + 8].float(), wz, padding=1), wz, dy[n0:n0 + 8].float())
        ref += gw
    err = (dw.float().cpu() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2e-3, float(err)                                 # fp32 accumulation of exact bf16 products: summation order only


@pytest.mark.parametrize('shape', [(64, 512, 512, 16, 16), (64, 512, 256, 32, 32), (64, 64, 32, 256, 256)],
                         ids=lambda s: f'N{s[0]}_{s[1]}to{s[2]}_{s[3]}x{s[4]}')
def test_modulated_weight_gradient_at_bench_shapes_vs_cpu_and_bitwise_reproducible(shape):
    """The generator's weight gradients carry per-image scales on both operands: blocks that stay inside one image scale their partial
    sums (256x256 maps), blocks that span images scale the fragments in registers (16x16 / 32x32 maps at batch 64).  The two-stage
    split-K combine has no atomics: two launches give the same bits."""
    from animeface_amd.implementations.StyleGAN2.conv import conv2d_wgrad_raw
    N, Cin, Cout, H, W = shape
    x, _, _ = _mk(N, Cin, Cout, H, W, 5)
    g = torch.Generator().manual_seed(6)
    dy = torch.randn(N, Cout, H, W, generator=g).to(torch.bfloat16)
    s_in, s_out = torch.rand(N, Cin, generator=g) + 0.5, torch.rand(N, Cout, generator=g) + 0.5
    xd, dyd = x.to(DEV).contiguous(memory_format=torch.channels_last), dy.to(DEV).contiguous(memory_format=torch.channels_last)
    dw = conv2d_wgrad_raw(xd, dyd, 3, in_scale=s_in.to(DEV), out_scale=s_out.to(DEV))
    dw2 = conv2d_wgrad_raw(xd, dyd, 3, in_scale=s_in.to(DEV), out_scale=s_out.to(DEV))
    assert torch.equal(dw, dw2)
    wz = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    ref = torch.zeros(Cout, Cin, 3, 3)
    for n0 in range(0, N, 8):
        xs = x[n0:n0 + 8].float() * s_in[n0:n0 + 8, :, None, None]
        ds = dy[n0:n0 + 8].float() * s_out[n0:n0 + 8, :, None, None]
        (gw,) = torch.autograd.grad(F.conv2d(xs, wz, padding=1), wz, ds)
        ref += gw
    err = (dw.float().cpu() - ref).abs().max() / ref.abs().max()
    assert float(err) < 8e-3, float(err)             # the in-register path rounds the scaled operands to bf16, as the forward conv does
