import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


@pytest.fixture
def wide_dispatch(request):
    """`wide_dispatch` parametrised True: the 128 x 512-tile ("wide") GEMM / fp16 weight-gradient / fp16x2 data-gradient kernels and the
    fused GEGLU epilogue -- what bench.py's B = 16 headline dispatches (ops.WIDE_MIN_M = 16384 rows) -- are taken from 1024 rows on, so
    that the B = 1 / 2 reference fixtures run through the headline's kernels; restored afterwards."""
    on = bool(getattr(request, 'param', False))
    if not on:
        yield False
        return
    from voxactb_amd import ops
    old = ops.WIDE_MIN_M
    ops.set_wide_min_rows(1024)
    try:
        yield True
    finally:
        ops.set_wide_min_rows(old)
