"""Adam over the flat parameter arena -- the reference's `optimizer: adam` alternative (agent :263-268:
`torch.optim.Adam(params, lr=lr, weight_decay=lambda_weight_l2)`), one fused launch instead of a foreach chain."""
import torch
from torch.optim import Optimizer

from ..._lib import VoxactbHipError, call


class Adam(Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._arena = None
        self.steps = 0
        self.skip_flag = None          # optional device int32: a negative value turns step() into a no-op on the device

    def attach(self, arena):
        self._arena = arena
        self.exp_avg = torch.zeros_like(arena.flat_w)
        self.exp_avg_sq = torch.zeros_like(arena.flat_w)
        for p, (off, n) in zip(arena.params, arena.segments):
            self.state[p]['exp_avg'] = self.exp_avg[off:off + n].view_as(p)
            self.state[p]['exp_avg_sq'] = self.exp_avg_sq[off:off + n].view_as(p)
            self.state[p]['step'] = 0

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._arena is None:
            raise VoxactbHipError('Adam.step(): parameters are not in a FlatParams arena (call attach())')
        if len(self.param_groups) != 1:
            raise VoxactbHipError('Adam: the fused step covers exactly one parameter group')
        g, a = self.param_groups[0], self._arena
        self.steps += 1
        call('vxb_adam_step_f32', a.flat_w, a.flat_g, self.exp_avg, self.exp_avg_sq, a.total, float(g['lr']), float(g['betas'][0]),
             float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), self.steps, self.skip_flag)
        for p in a.params:
            self.state[p]['step'] = self.steps
        return loss
