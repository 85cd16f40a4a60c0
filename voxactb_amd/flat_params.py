"""One contiguous fp32 arena for all parameters and one for all gradients.

The reference lets DDP bucket 118 gradient tensors (25 MB buckets over gloo, qattention_peract_bc_agent.py:50-54)
and LAMB loop over them one by one.  Here every Parameter's `.data` and `.grad` are views into two flat buffers, so
the data-parallel exchange is a handful of in-place RCCL all-reduces over contiguous slices of ONE buffer and the
optimizer is one fused launch.  `named_parameters()`, `state_dict()`, `load_state_dict()`, `param.grad`
(update_summaries, agent :814-821) keep working.

Overlap with the backward pass: the arena follows the module's registration order, and the backward pass finishes
gradients from the END of that order towards its start (heads, decoder, the self-attention layers 5 .. 0, then the
input side).  `bucket(prefixes)` names a contiguous slice by parameter-name prefixes; the engine calls
`reduce_bucket(name)` as soon as the last kernel writing into that slice has been enqueued, which starts an
asynchronous all-reduce (RCCL runs it on its own stream behind an event on the compute stream) while the backward
pass continues; `finish_reduce()` makes the compute stream wait for all of them before the optimizer.
"""
import os

import torch

ALIGN = 64        # floats: every tensor starts on a 256-byte boundary
# debug / test switch: issue the collectives even with a single rank (RCCL refuses two ranks on one GPU, so a 1-GPU box can
# only exercise the real backend -- communicator creation, the asynchronous bucket all-reduces on RCCL's stream behind the
# compute stream's event, the wait before the optimizer -- with world size 1; tests/test_distributed_gpu.py)
FORCE_COLLECTIVES = os.environ.get('VOXACTB_FORCE_COLLECTIVES', '0') == '1'


class FlatParams:
    def __init__(self, module: torch.nn.Module, device):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.names = [n for n, p in module.named_parameters() if p.requires_grad]
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.flat_w = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=device)
        self.segments = []
        for p, off in zip(self.params, offs):
            n = p.numel()
            self.flat_w[off:off + n].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = self.flat_w[off:off + n].view(p.shape)
            p.grad = self.flat_g[off:off + n].view(p.shape)
            self.segments.append((off, n))
        self.total = total

        self._buckets = {}
        self._pending = []
        self._done = set()
        # exchange timing (bench.py at N > 1): per step, events on the compute stream at each bucket's start, in front of and behind
        # the waits of finish_reduce -> how long after its start a bucket was complete at the latest, and how long the compute
        # stream stood still for the exchange (the exposed part)
        self.timing = False
        self.timing_records = []
        self._t_start = {}

    def zero_grad(self):
        self.flat_g.zero_()
        self._done.clear()

    # ------------------------------------------------------------------ buckets of the gradient exchange
    def bucket(self, name, prefixes):
        """Declare the slice of the flat buffers that holds every parameter whose name starts with one of `prefixes`
        (must be contiguous in registration order).  Returns (first float, end float)."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(tuple(prefixes))]
        if not idx:
            raise ValueError('no parameter matches %r' % (prefixes,))
        if idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError('parameters matching %r are not contiguous in the arena' % (prefixes,))
        lo = self.segments[idx[0]][0]
        hi = self.total if idx[-1] + 1 == len(self.segments) else self.segments[idx[-1] + 1][0]
        self._buckets[name] = (lo, hi)
        return lo, hi

    def buckets_cover_everything(self):
        spans = sorted(self._buckets.values())
        return bool(spans) and spans[0][0] == 0 and spans[-1][1] == self.total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))

    @staticmethod
    def _initialised():
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()

    @staticmethod
    def _world():
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def reduce_bucket(self, name):
        """Start the sum over ranks of one bucket (no-op for a single rank).  The 1/world factor is folded into the loss
        scale, see QAttentionPerActBCAgent.update."""
        if name in self._done:
            raise RuntimeError('bucket %r reduced twice in one step' % name)
        self._done.add(name)
        if self._world() > 1 or (FORCE_COLLECTIVES and self._initialised()):
            import torch.distributed as dist
            lo, hi = self._buckets[name]
            if self.timing:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                self._t_start[name] = ev
            self._pending.append(dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish_reduce(self):
        """Every bucket that has not been started yet goes out now; then the compute stream waits for all of them."""
        for name in self._buckets:
            if name not in self._done:
                self.reduce_bucket(name)
        if not self._buckets:
            self.all_reduce_grads()
        if self.timing and self._pending:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for w in self._pending:
                w.wait()
            e1.record()
            self.timing_records.append((dict(self._t_start), e0, e1))
            self._t_start = {}
        else:
            for w in self._pending:
                w.wait()
        self._pending = []

    def exchange_summary(self):
        """(call after torch.cuda.synchronize()) -> dict for the bench JSON: bucket sizes, mean exposed wait per step and, per bucket, the
        mean time from its start to the end of the step's waits (an upper bound of its all-reduce time)."""
        n = len(self.timing_records)
        out = {'buckets_bytes': {k: 4 * (hi - lo) for k, (lo, hi) in self._buckets.items()}, 'steps_timed': n}
        if n:
            out['exposed_wait_ms_per_step'] = sum(e0.elapsed_time(e1) for _, e0, e1 in self.timing_records) / n
            names = list(self.timing_records[0][0])
            out['start_to_done_ms'] = {k: sum(r[0][k].elapsed_time(r[2]) for r in self.timing_records if k in r[0]) / n for k in names}
        self.timing_records = []
        return out

    def broadcast_weights(self, src=0):
        """Rank `src`'s parameters to every rank -- what DistributedDataParallel does implicitly when it wraps the module
        (agent :50-54): run_seed_fn.py never seeds torch, so without it every replica would start from its own init."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES):
            dist.broadcast(self.flat_w, src=src)

    def all_reduce_grads(self):
        """sum over ranks (the 1/world factor is folded into the loss scale, see QAttentionPerActBCAgent.update)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM)
