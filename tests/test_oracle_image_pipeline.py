"""The input-transform oracle against Pillow's own outputs (fixture written by tools/make_golden.py) -- and against Pillow itself when it
is importable.  CPU only."""
import numpy as np
import pytest

from oracle import image_pipeline as P


def test_resample_restatement_is_bit_exact_with_pillow_fixture(golden):
    g = golden('image_pipeline')
    for idx, (H, W, size, oh, ow) in enumerate(g['cases'].tolist()):
        assert P.resized_shape(H, W, size) == (oh, ow)
        got = P.resample_bilinear_u8(g[f'in{idx}'], oh, ow)
        assert got.dtype == np.uint8 and np.array_equal(got, g[f'out{idx}']), idx


def test_resample_restatement_against_live_pillow():
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(3)
    for (H, W, oh, ow) in [(61, 47, 30, 23), (30, 40, 77, 99), (128, 128, 64, 64), (200, 120, 128, 76)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(P.resample_bilinear_u8(img, oh, ow), ref)


def test_default_transform_shapes_range_and_flip():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (90, 160, 3), dtype=np.uint8)
    a = P.default_transform(img, 64, flip=False)
    b = P.default_transform(img, 64, flip=True)
    assert a.shape == (3, 64, 64) and a.dtype == np.float32 and -1.0 <= a.min() and a.max() <= 1.0
    assert np.array_equal(a[:, :, ::-1], b)
    # Resize(64) of 90x160 -> 64x113, centre crop origin (0, round(24.5)) = (0, 24)
    assert P.resized_shape(90, 160, 64) == (64, 113) and P.center_crop_origin(64, 113, 64) == (0, 24)
