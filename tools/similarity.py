"""Line-level similarity of a product file to its reference counterpart (whitespace / comment-normalised lines longer than 12 characters that
exist verbatim in the other file) -- the check the round-1 review used.  Container only (/root/reference is not shipped).
    python tools/similarity.py"""
import re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [('animeface_amd/stylegan3_ops/conv2d_resample.py', 'thirdparty/stylegan3_ops/ops/conv2d_resample.py'),
         ('animeface_amd/implementations/StyleGAN3/model.py', 'implementations/StyleGAN3/model.py'),
         ('animeface_amd/implementations/StyleGAN2/model.py', 'implementations/StyleGAN2/model.py'),
         ('animeface_amd/stylegan3_ops/upfirdn2d.py', 'thirdparty/stylegan3_ops/ops/upfirdn2d.py'),
         ('animeface_amd/stylegan3_ops/filtered_lrelu.py', 'thirdparty/stylegan3_ops/ops/filtered_lrelu.py'),
         ('animeface_amd/stylegan3_ops/bias_act.py', 'thirdparty/stylegan3_ops/ops/bias_act.py')]
def norm(path):
    out = []
    for line in open(path, errors='replace'):
        line = re.sub(r'#.*$', '', line)
        line = re.sub(r'\s+', '', line)
        if len(line) > 12: out.append(line)
    return out
for mine, ref in PAIRS:
    a, b = norm(os.path.join(ROOT, mine)), set(norm(os.path.join('/root/reference', ref)))
    same = sum(1 for l in a if l in b)
    print(f'{same:4d} / {len(a):4d} normalised lines of {mine} exist verbatim in {ref}')
