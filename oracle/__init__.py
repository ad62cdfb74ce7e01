"""CPU oracle for the StyleGAN2/3 hot path  --  TEST INFRASTRUCTURE ONLY.

Every function here is a plain-PyTorch (CPU) restatement of the reference's
algorithm for the hot path named by BASELINE.json, each citing the reference
file:line it follows.  Nothing under ``animeface_amd/`` (the product) may
import this package: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, generated in the
build container by ``tools/make_golden.py`` (imports /root/reference on CPU)
and committed as fixtures under ``tests/golden/``.  ``tests/test_oracle_*.py``
replays every fixture through this package.
"""
