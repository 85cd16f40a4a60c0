import sys; sys.path.insert(0,'/root/repo')
import numpy as np, os
from tests.test_c2_reference_gpu import _run
from tests.conftest import GOLDEN
for fx in ['f3v_encoder_tiny_iterations2', 'f3v_encoder_c1_iterations3', 'f3v_encoder_c1_no_language', 'f3v_encoder_c1_no_skip_connection', 'f3v_encoder_c1_no_perceiver', 'f3v_encoder_c1_pos_encoding_grid_only', 'f3v_encoder_c1_lang_concat', 'f3v_encoder_c1_weight_tie_layers']:
    g = np.load(os.path.join(GOLDEN, fx + '.npz'), allow_pickle=False)
    try:
        _run(g, 'bf16x3', fx[4:], backward=True)
        print('PASS', fx)
    except AssertionError as e:
        print('FAIL', fx, str(e)[:600])
