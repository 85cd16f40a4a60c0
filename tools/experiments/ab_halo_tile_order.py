#!/usr/bin/env python
"""Same-box A/B of the LDS-halo convs' tile order on the whole training step: VXB_HALO_DBG=128 (experiment bit 0x80: row order) vs 0
(4 x 4 x 4 blocks of tiles, shipped).   for d in 128 0 128 0; do VXB_HALO_DBG=$d python tools/experiments/ab_halo_tile_order.py; done"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from voxactb_amd import _lib  # noqa: E402

_lib.lib().vxb_debug_set_halo_experiment(int(os.environ.get('VXB_HALO_DBG', '0')))
sys.argv = ['bench.py', '--steps', '8', '--warmup', '2', '--no-cpu-baseline', '--no-other-modes']
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
