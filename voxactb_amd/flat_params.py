"""One contiguous fp32 arena for all parameters and one for all gradients.

The reference lets DDP bucket 118 gradient tensors (25 MB buckets over gloo, qattention_peract_bc_agent.py:50-54)
and LAMB loop over them one by one.  Here every Parameter's `.data` and `.grad` are views into two flat buffers, so
the data-parallel exchange is ONE in-place RCCL all-reduce over xGMI and the optimizer is one fused launch.
`named_parameters()`, `state_dict()`, `load_state_dict()`, `param.grad` (update_summaries, agent :814-821) keep working.
"""
import torch

ALIGN = 64   # floats: every tensor starts on a 256-byte boundary


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

    def zero_grad(self):
        self.flat_g.zero_()

    def broadcast_weights(self, src=0):
        """Rank `src`'s parameters to every rank -- what DistributedDataParallel does implicitly when it wraps the module
        (agent :50-54): run_seed_fn.py never seeds torch, so without it every replica would start from its own init."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.flat_w, src=src)

    def all_reduce_grads(self):
        """sum over ranks (the 1/world factor is folded into the loss scale, see QAttentionPerActBCAgent.update)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM)
