"""Lamb -- drop-in for the reference optimizer (reference: peract/helpers/optim/lamb.py:12-124).

Same constructor / `step()` / `state` layout semantics (per-parameter `exp_avg`, `exp_avg_sq`, `step`,
`weight_norm`-style trust ratio, no bias correction, ||w|| clamped to [0, 10]), but ONE fused multi-tensor
launch sequence (3 kernels, no host sync) over flat buffers instead of 118 x (2 reductions + a
tensor->bool sync).  Parameters must live in a `FlatParams` arena (see flat_params.py); the agent arranges that.
Every arena parameter has a gradient view (zero when nothing flowed into it), so none is skipped the way the reference
skips `p.grad is None` (lamb.py:75-76); on this path the engine writes a gradient for every parameter it declares.
"""
import numpy as np
import torch
from torch.optim import Optimizer

from ..._lib import call, VoxactbHipError

CHUNK = 4096


class Lamb(Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0, adam=False):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        if adam:
            raise NotImplementedError('adam=True (trust ratio forced to 1) is a debugging switch upstream; not built')
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.adam = adam
        super(Lamb, self).__init__(params, defaults)
        self._arena = None
        self._plan = None
        self.skip_flag = None          # optional device int32: a negative value turns step() into a no-op on the device

    def attach(self, arena):
        """arena: FlatParams holding every parameter of this optimizer (contiguous flat weights / grads)."""
        self._arena = arena
        dev = arena.flat_w.device
        self.exp_avg = torch.zeros_like(arena.flat_w)
        self.exp_avg_sq = torch.zeros_like(arena.flat_w)
        self._upd = torch.empty_like(arena.flat_w)
        chunks, first = [], [0]
        for t, (off, n) in enumerate(arena.segments):
            for s in range(0, n, CHUNK):
                chunks.append((t, off + s, min(CHUNK, n - s)))
            first.append(len(chunks))
        self._nchunks, self._ntensors = len(chunks), len(arena.segments)
        self._chunks = torch.tensor(np.array(chunks, np.int32).reshape(-1), dtype=torch.int32, device=dev)
        self._first = torch.tensor(first, dtype=torch.int32, device=dev)
        self._part = torch.empty(2 * self._nchunks, dtype=torch.float32, device=dev)
        self.trust_ratio = torch.ones(self._ntensors, dtype=torch.float32, device=dev)
        self.steps = 0
        for p, (off, n) in zip(arena.params, arena.segments):   # state views, as the reference keeps per-parameter state
            self.state[p]['exp_avg'] = self.exp_avg[off:off + n].view_as(p)
            self.state[p]['exp_avg_sq'] = self.exp_avg_sq[off:off + n].view_as(p)
            self.state[p]['step'] = 0

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._arena is None:
            raise VoxactbHipError('Lamb.step(): parameters are not in a FlatParams arena (call attach())')
        if len(self.param_groups) != 1:
            raise VoxactbHipError('Lamb: the fused step covers exactly one parameter group (the agent creates one, agent :243-249)')
        g = self.param_groups[0]
        a = self._arena
        call('vxb_lamb_step_f32', a.flat_w, a.flat_g, self.exp_avg, self.exp_avg_sq, self._upd, self._chunks, self._nchunks,
             self._first, self._ntensors, self._part, self.trust_ratio, float(g['lr']), float(g['betas'][0]),
             float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), self.skip_flag)
        self.steps += 1
        for p in a.params:                   # per-parameter step counter, as the reference keeps it (lamb.py:90)
            self.state[p]['step'] = self.steps
        return loss
