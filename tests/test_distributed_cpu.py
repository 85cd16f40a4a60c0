"""CPU, world_size 2 over gloo: the data-parallel exchange of the hot path (FlatParams.all_reduce_grads) and the
per-rank replay sharding arithmetic.  The GPU run uses the same code with backend "nccl" (= RCCL over xGMI)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from voxactb_amd.flat_params import FlatParams
    torch.manual_seed(0)                               # same init on every rank
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.LayerNorm(5), torch.nn.Linear(5, 3))
    arena = FlatParams(net, 'cpu')
    # parameters / grads are views into the flat buffers
    assert all(p.data.data_ptr() >= arena.flat_w.data_ptr() for p in net.parameters())
    B = 4
    torch.manual_seed(100 + rank)                      # different shard per rank
    x, y = torch.randn(B, 7), torch.randn(B, 3)
    arena.zero_grad()
    # loss scale 1/(B*world) as QAttentionPerActBCAgent.update uses -> SUM all-reduce == DDP's mean over ranks
    loss = ((net(x) - y) ** 2).sum() / (B * world)
    grads = torch.autograd.grad(loss, list(net.parameters()))
    for p, g in zip(net.parameters(), grads):
        p.grad.copy_(g)
    local = arena.flat_g.clone()
    arena.all_reduce_grads()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(arena.flat_g, sum(gathered), atol=1e-7)
    # every rank now holds identical gradients -> identical LAMB update -> weights stay in sync without a broadcast
    ref = [torch.zeros_like(arena.flat_g) for _ in range(world)]
    dist.all_gather(ref, arena.flat_g)
    assert torch.equal(ref[0], ref[1])
    q.put((rank, float(arena.flat_g.abs().sum())))
    dist.destroy_process_group()


def test_grad_allreduce_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get() for _ in range(world))
    assert abs(res[0] - res[1]) < 1e-6


def test_rank_sharded_batches_differ_and_schema():
    """bench.py gives rank r the synthetic shard seeded with 100*r (weak scaling); shards must differ, schema must
    be the one launch_utils.create_replay declares (reference launch_utils.py:56-145)."""
    from voxactb_amd import synthetic
    from voxactb_amd.agents.peract_bc import launch_utils as lu
    a = synthetic.make_replay_sample(2, ['front', 'wrist'], (8, 8), 16, 4, seed=0)
    b = synthetic.make_replay_sample(2, ['front', 'wrist'], (8, 8), 16, 4, seed=100)
    assert not torch.equal(a['front_point_cloud'], b['front_point_cloud'])
    for name, shape, dt in lu.replay_schema(['front', 'wrist'], [16], (8, 8)):
        if name in ('task', 'lang_goal'):
            continue
        assert tuple(a[name].shape) == (2, 1) + tuple(shape), name


def test_reference_replay_index_sharding():
    """task_uniform_replay_buffer.py:103-108: rank r samples task_idxs[r : total : num_replicas]."""
    idxs = list(range(23))
    shards = [idxs[r::4] for r in range(4)]
    assert sorted(sum(shards, [])) == idxs and all(len(set(s) & set(t)) == 0 for i, s in enumerate(shards) for t in shards[i + 1:])
