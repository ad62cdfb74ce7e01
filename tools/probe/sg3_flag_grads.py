"""The layer-by-layer bf16 StyleGAN3 test (tests/test_hip_sg3.py, fixture configuration) under the four RES_FUSED x FUSED_SCALARS combinations:
prints the relative rms error of the sampled parameter gradients against the bf16-emulating oracle.  python tools/probe/sg3_flag_grads.py"""
import sys, os, itertools, io, contextlib
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); os.chdir(root)
import pytest
from animeface_amd.implementations.StyleGAN3 import model as M
import oracle.stylegan3 as S3
for res, fs in itertools.product([True, False], [True, False]):
    M.RES_FUSED, M.FUSED_SCALARS = res, fs
    S3.RESBLOCK_SUM_IN_SKIP_CONV = res
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = pytest.main(['tests/test_hip_sg3.py', '-q', '-s', '-x', '-k', 'bf16_networks_layer and fixture', '-p', 'no:cacheprovider'])
    lines = [l for l in buf.getvalue().splitlines() if 'grad ' in l and 'rms' in l]
    print('RES_FUSED', res, 'FUSED_SCALARS', fs, 'rc', int(rc))
    print('   ' + ' | '.join(l.strip().replace('grad ', '').replace(': rms rel ', '=') for l in lines))
