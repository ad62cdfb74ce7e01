#!/usr/bin/env python3
"""The snippet of the reference's weights.md (:10-40) on this package: build the 128-pixel generator, load a checkpoint in the reference's
format (``G_ema.state_dict()`` saved with torch.save), generate images and a style-mixed image.

    python tools/load_weights_md.py [StyleGAN2_animeface_128pix.pt] [--out images.pt]

Without a file (the published checkpoint is a download; this environment has no network) a randomly initialised generator's own
state_dict is saved and loaded back, which exercises the same code path and file format."""
import argparse
import functools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animeface_amd.implementations.StyleGAN2.model import Generator, init_weight_N01     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('checkpoint', nargs='?')
ap.add_argument('--out', default=None)
ap.add_argument('--num-images', type=int, default=4)
a = ap.parse_args()
dev = torch.device('cuda')
G = Generator(image_size=128, image_channels=3, style_dim=512, channels=32, max_channels=512, block_num_conv=2,
              map_num_layers=8, map_lr=0.01).to(dev)
if a.checkpoint is None:
    torch.manual_seed(0)
    G.init_weight(functools.partial(init_weight_N01, lr=0.01), init_weight_N01)
    a.checkpoint = '/tmp/StyleGAN2_random_128pix.pt'
    torch.save(G.state_dict(), a.checkpoint)
    print('no checkpoint given: wrote a randomly initialised one to', a.checkpoint)
state_dict = torch.load(a.checkpoint, map_location=dev)
G.load_state_dict(state_dict)                 # strict: the key set is the reference's (tests/test_abi.py pins it)
G.eval()


def sampler(num_image):
    return torch.randn(num_image, 512, device=dev)


with torch.no_grad():
    images, _ = G(sampler(a.num_images))
    mixed, _ = G((sampler(a.num_images), sampler(a.num_images)), injection=4)
print('images', tuple(images.shape), 'range', float(images.min()), float(images.max()), '| style-mixed', tuple(mixed.shape))
if a.out:
    torch.save(dict(images=images.cpu(), mixed=mixed.cpu()), a.out)
