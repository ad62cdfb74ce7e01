"""The GPU input transform (animeface_amd/dataset.py, agf_image_resample_rows / agf_image_finish) against Pillow's own BILINEAR outputs
(fixture: tests/golden/image_pipeline.npz) and the oracle's restatement of the whole torchvision transform chain; integer work: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import image_pipeline as P

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_resize_crop_is_bit_exact_with_pillow(golden):
    from animeface_amd.dataset import GpuTransform
    g = golden('image_pipeline')
    for idx, (H, W, size, oh, ow) in enumerate(g['cases'].tolist()):
        S = min(oh, ow)                                        # Resize(size) -> CenterCrop(S): the crop of Pillow's resized image
        tf = GpuTransform(S, resize_scale=size / S, hflip=False, normalize=False)
        assert tf.resize_to == size
        img = torch.from_numpy(g[f'in{idx}']).to(DEV)
        out = tf(torch.stack([img, img]))
        top, left = P.center_crop_origin(oh, ow, S)
        ref = g[f'out{idx}'][top:top + S, left:left + S].astype(np.float32).transpose(2, 0, 1) / np.float32(255)
        got = out.cpu().numpy()
        assert got.shape == (2, 3, S, S)
        assert np.array_equal(got[0], ref) and np.array_equal(got[1], ref), (idx, np.abs(got[0] - ref).max())


@pytest.mark.parametrize('shape,size', [((5, 90, 160, 3), 64), ((3, 200, 130, 3), 128), ((4, 64, 64, 3), 64), ((2, 70, 50, 3), 48)])
def test_default_transform_matches_the_oracle_chain(shape, size):
    from animeface_amd.dataset import make_default_transform
    from animeface_amd import rng
    g = np.random.default_rng(11)
    batch = g.integers(0, 256, shape, dtype=np.uint8)
    tf = make_default_transform(size)
    with rng.cpu_stream():
        torch.manual_seed(3)
        out = tf(torch.from_numpy(batch).to(DEV))
        torch.manual_seed(3)
        flips = [bool(torch.rand(1) < 0.5) for _ in range(shape[0])]          # RandomHorizontalFlip's draw, per image, in order
    assert any(flips) or shape[0] < 3
    ref = np.stack([P.default_transform(batch[i], size, flip=flips[i]) for i in range(shape[0])])
    got = out.cpu().numpy()
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert np.array_equal(got, ref), np.abs(got - ref).max()
    # bf16 output: the same values rounded once
    out16 = type(tf)(size, dtype=torch.bfloat16)(torch.from_numpy(batch).to(DEV), flips=flips)
    assert torch.equal(out16.cpu(), torch.from_numpy(ref).to(torch.bfloat16))


def test_full_size_batch_properties():
    """64 images of 300 x 300 -> 256 x 256: flips are involutions, a constant image stays constant, output in [-1, 1]."""
    from animeface_amd.dataset import GpuTransform
    x = torch.randint(0, 256, (64, 300, 300, 3), dtype=torch.uint8, device=DEV)
    tf = GpuTransform(256)
    a = tf(x, flips=[False] * 64)
    b = tf(x, flips=[True] * 64)
    assert torch.equal(a.flip(3), b) and float(a.min()) >= -1 and float(a.max()) <= 1
    c = tf(torch.full((2, 300, 300, 3), 200, dtype=torch.uint8, device=DEV), flips=[False, True])
    assert torch.equal(c, torch.full_like(c, (200 / 255 - 0.5) / 0.5))
