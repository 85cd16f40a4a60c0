import sys; sys.path.insert(0,'/root/repo')
import numpy as np, os
from tests.test_c2_reference_gpu import _run
from tests.conftest import GOLDEN
for fx in ['f5v50a_encoder_release_digest', 'f5v50b_encoder_release_digest']:
    g = np.load(os.path.join(GOLDEN, fx + '.npz'), allow_pickle=False)
    try:
        _run(g, 'bf16x3', fx[:6], backward=True)
        print('PASS', fx)
    except AssertionError as e:
        print('FAIL', fx, str(e)[:900])
