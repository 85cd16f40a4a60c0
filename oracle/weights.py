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
"""
import zlib
import numpy as np
import torch

LRELU_SLOPE = 0.02  # network_utils.py:12


def _rng(name: str, seed: int):
    key = zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))


def hashed_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    g = _rng(name, seed)
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = name.split('.')[-1]
    if name in ('latents', 'pos_encoding'):
        a = g.standard_normal(n)
    elif '.norm' in name or name.startswith('norm'):
        u = g.uniform(-1.0, 1.0, n)
        a = (1.0 + 0.1 * u) if leaf == 'weight' else 0.1 * u
    elif leaf == 'weight':
        fan_in = int(np.prod(shape[1:]))
        if 'conv3d' in name or name.endswith('linear.weight'):
            bound = np.sqrt(6.0 / ((1.0 + LRELU_SLOPE ** 2) * fan_in))
        else:
            bound = 1.0 / np.sqrt(fan_in)
        a = g.uniform(-bound, bound, n)
    elif leaf == 'bias':
        a = 0.05 * g.uniform(-1.0, 1.0, n)
    else:
        a = g.uniform(-1.0, 1.0, n)
    return torch.from_numpy(a.astype(np.float32).reshape(shape))


def hashed_state_dict(shapes: dict, seed: int = 0) -> dict:
    """shapes: {param_name: shape}.  Returns {param_name: fp32 tensor}."""
    return {k: hashed_tensor(k, v, seed) for k, v in shapes.items()}


def hashed_uniform(name: str, shape, lo=0.0, hi=1.0, seed: int = 0) -> torch.Tensor:
    g = _rng('u:' + name, seed)
    a = g.uniform(lo, hi, int(np.prod(shape)))
    return torch.from_numpy(a.astype(np.float32).reshape(tuple(shape)))


def hashed_normal(name: str, shape, seed: int = 0) -> torch.Tensor:
    g = _rng('n:' + name, seed)
    a = g.standard_normal(int(np.prod(shape)))
    return torch.from_numpy(a.astype(np.float32).reshape(tuple(shape)))


def hashed_int(name: str, shape, lo: int, hi: int, seed: int = 0) -> torch.Tensor:
    """integers in [lo, hi)."""
    g = _rng('i:' + name, seed)
    a = g.integers(lo, hi, int(np.prod(shape)))
    return torch.from_numpy(a.astype(np.int64).reshape(tuple(shape)))
