"""Deterministic name-hashed parameter generator (test infrastructure).

The reference initialises parameters from torch's global RNG
(perceiver_lang_io.py:197-205,234; network_utils.py:140-154,263-276).  Golden
fixtures cannot ship 133 MB of weights, so both the fixture generator (which
loads them into the *reference* module) and the tests (which load them into the
product module and the oracle) derive every tensor from
`(parameter name, seed)` with numpy's Philox counter RNG.  Magnitudes follow
the reference's init families so activations stay O(1):
  conv / dense with lrelu : kaiming-uniform bound sqrt(6 / ((1+a^2) fan_in))
  plain nn.Linear          : bound 1/sqrt(fan_in)
  LayerNorm                : weight 1 + 0.1 u, bias 0.1 u   (perturbed on purpose
                             so affine paths are exercised)
  biases                   : 0.1 u / sqrt(fan_in)-ish small non-zero values
  latents / pos_encoding   : N(0, 1)

The generator itself lives with the other synthetic-data code in `voxactb_amd/synthetic.py` (bench.py's reference-digest
check needs it too and must not import `oracle/`); this module re-exports it under the names the tests use.
"""
from voxactb_amd.synthetic import (LRELU_SLOPE, hashed_int, hashed_normal, hashed_state_dict, hashed_tensor,  # noqa: F401
                                   hashed_uniform, project, projection_error, projection_signs)
