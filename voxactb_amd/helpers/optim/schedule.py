"""Cosine learning-rate schedule with warm-up and hard restarts -- what the reference builds with
`transformers.get_cosine_with_hard_restarts_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps,
num_cycles=training_iterations // 10000)` when `method.lr_scheduler` is on (agent :273-279, stepped at :595-597)."""
import math


class CosineWithHardRestarts:
    """lr(step) = base_lr * lambda(step):  step / warmup during warm-up, then 0.5 (1 + cos(pi ((cycles * progress) mod 1)))
    with progress = (step - warmup) / (total - warmup), 0 once progress reaches 1.  `step()` after every optimizer step."""

    def __init__(self, optimizer, num_warmup_steps, num_training_steps, num_cycles=1):
        self.optimizer = optimizer
        self.warmup, self.total, self.cycles = int(num_warmup_steps), int(num_training_steps), num_cycles
        self.base_lrs = [g['lr'] for g in optimizer.param_groups]
        self.last_epoch = 0
        self._apply()

    def factor(self, step):
        if step < self.warmup:
            return float(step) / float(max(1, self.warmup))
        progress = float(step - self.warmup) / float(max(1, self.total - self.warmup))
        if progress >= 1.0:
            return 0.0
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(self.cycles) * progress) % 1.0))))

    def _apply(self):
        f = self.factor(self.last_epoch)
        self._last_lr = [b * f for b in self.base_lrs]
        for g, lr in zip(self.optimizer.param_groups, self._last_lr):
            g['lr'] = lr

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return self._last_lr

    def state_dict(self):
        return {'last_epoch': self.last_epoch, 'base_lrs': self.base_lrs}

    def load_state_dict(self, sd):
        self.last_epoch, self.base_lrs = sd['last_epoch'], sd['base_lrs']
        self._apply()
