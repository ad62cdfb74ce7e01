#!/usr/bin/env python3
"""bench.py on a probe build of the library (tools/probe/build_variant.sh):  AGF_PROBE_LIB=<name> python tools/probe/bench_with_lib.py <bench.py arguments>"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from animeface_amd import _lib
if os.environ.get('AGF_PROBE_LIB'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libagf_ops_%s.so' % os.environ['AGF_PROBE_LIB'])
import bench
bench.main()
