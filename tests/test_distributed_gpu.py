"""GPU, two ranks on ONE device over gloo (RCCL refuses two ranks per GPU; the code path is the same `torch.distributed`
calls the 8-GPU run makes with backend "nccl"): the real `update()` step under data parallelism.

  (a) ranks that initialise their networks from DIFFERENT seeds hold bit-identical weights after build() (rank 0's are
      broadcast, as DistributedDataParallel does when it wraps the module, reference agent :50-54) and after two update()s
      on different shards;
  (b) the exchanged gradient equals the gradient of ONE process on the concatenated batch (loss = mean over the global batch,
      agent :578), and so does the LAMB step taken from it;
  (c) every bucket of the overlapped exchange is reduced exactly once per step.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(method__voxel_sizes=[16], method__voxel_patch_size=3, method__voxel_patch_stride=4, method__transformer_depth=2,
           method__num_latents=32, method__input_dropout=0.0, method__attn_dropout=0.0,
           rlbench__cameras=['front', 'wrist'], rlbench__camera_resolution=[16, 16])
B = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_agent(batch, seed):
    # exact-fp32 matrix cores: two batch compositions then differ by summation order only (~1e-6); the default bf16x3
    # split products add ~1e-4 of the largest gradient entry, which would blur check (b)
    os.environ['VOXACTB_PRECISION'] = 'fp32'
    from voxactb_amd.agents.peract_bc import launch_utils as lu
    cfg = lu.default_cfg(replay__batch_size=batch, **CFG)
    cfg.method.transform_augmentation.apply_se3 = False
    torch.manual_seed(seed)
    agent = lu.create_agent(cfg)
    agent.build(training=True, device=0)
    return agent, agent._pose_agent._qattention_agents[0]


def _shard(rank):
    from voxactb_amd import synthetic
    return synthetic.make_replay_sample(B, CFG['rlbench__cameras'], (16, 16), 16, 4, seed=50 + rank)


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    agent, qa = _make_agent(B, seed=1000 + 17 * rank)              # a different initialisation on every rank
    w_build = qa._arena.flat_w.clone()
    calls = []
    orig = qa._arena.reduce_bucket
    qa._arena.reduce_bucket = lambda name: (calls.append(name), orig(name))[1]
    losses = []
    grads = None
    for step in range(2):
        rs = {k: v.to('cuda:0') for k, v in _shard(rank if step == 0 else 1 - rank).items()}
        losses.append(float(agent.update(step, rs)['total_losses']))
        if step == 0:
            grads = qa._arena.flat_g.clone()
            w_step0 = qa._arena.flat_w.clone()
    torch.save(dict(w_build=w_build.cpu(), w_step0=w_step0.cpu(), w_final=qa._arena.flat_w.cpu(), grads=grads.cpu(), losses=losses,
                    calls=calls, buckets=list(qa._arena._buckets)), os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_update_matches_one_process_on_the_concatenated_batch(tmp_path, monkeypatch):
    monkeypatch.setenv('VOXACTB_PRECISION', 'fp32')
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    r0, r1 = (torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in range(world))
    # (a) one model: identical after build (although the ranks seeded differently) and after every step
    assert torch.equal(r0['w_build'], r1['w_build'])
    assert torch.equal(r0['w_step0'], r1['w_step0']) and torch.equal(r0['w_final'], r1['w_final'])
    assert torch.equal(r0['grads'], r1['grads'])                    # the summed gradient is the same tensor on both ranks
    assert not torch.equal(r0['w_build'], r0['w_final'])
    # (c) every bucket once per step, backward order: tail, layers high -> low, head
    assert r0['calls'] == r0['buckets'] * 2 and r0['buckets'][0] == 'tail' and r0['buckets'][-1] == 'head'
    # (b) one process, global batch = the two shards concatenated, starting from the broadcast weights
    agent, qa = _make_agent(2 * B, seed=1)
    qa._arena.flat_w.copy_(r0['w_build'].to('cuda:0'))
    s0, s1 = _shard(0), _shard(1)
    rs = {k: torch.cat([s0[k], s1[k]], 0).to('cuda:0') for k in s0}
    loss = float(agent.update(0, rs)['total_losses'])
    g1 = qa._arena.flat_g.cpu()
    den = float(g1.abs().max())
    assert float((g1 - r0['grads']).abs().max()) < 2e-5 * den + 1e-7, (float((g1 - r0['grads']).abs().max()), den)
    assert abs(loss - 0.5 * (r0['losses'][0] + r1['losses'][0])) < 1e-5          # mean over the global batch
    dw = (qa._arena.flat_w.cpu() - r0['w_step0']).abs().max()
    assert float(dw) < 5e-6, float(dw)


def _rccl_worker(port, out_dir, force):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['VOXACTB_FORCE_COLLECTIVES'] = '1' if force else '0'
    torch.cuda.set_device(0)
    if force:
        dist.init_process_group('nccl', rank=0, world_size=1)          # backend "nccl" IS RCCL on ROCm
    agent, qa = _make_agent(B, seed=7)
    calls = []
    orig = qa._arena.reduce_bucket
    qa._arena.reduce_bucket = lambda name: (calls.append((name, len(qa._arena._pending))), orig(name))[1]
    losses = []
    for step in range(3):
        rs = {k: v.to('cuda:0') for k, v in _shard(step % 2).items()}
        losses.append(float(agent.update(step, rs)['total_losses']))
    torch.cuda.synchronize()
    torch.save(dict(w=qa._arena.flat_w.cpu(), losses=losses, calls=[c[0] for c in calls], buckets=list(qa._arena._buckets)),
               os.path.join(out_dir, 'rccl%d.pt' % int(force)))
    if force:
        dist.barrier()
        dist.destroy_process_group()


def test_bucketed_exchange_on_the_real_rccl_backend(tmp_path):
    """The exchange the 8-GPU run makes -- rank-0 broadcast at build(), one asynchronous in-place all-reduce per gradient bucket
    started from inside the backward pass, the wait before LAMB -- on the RCCL backend itself.  One GPU means world size 1 (the
    collectives are identities), so three update() steps must leave exactly the weights of a run without any process group."""
    ctx = mp.get_context('spawn')
    for force in (True, False):
        p = ctx.Process(target=_rccl_worker, args=(_free_port(), str(tmp_path), force))
        p.start()
        p.join(600)
        assert p.exitcode == 0, 'worker (collectives %s) failed' % force
    a, b = (torch.load(os.path.join(str(tmp_path), 'rccl%d.pt' % f)) for f in (1, 0))
    assert a['calls'] == a['buckets'] * 3 and a['buckets'][0] == 'tail' and a['buckets'][-1] == 'head'
    assert a['losses'] == b['losses']
    assert torch.equal(a['w'], b['w'])


# ------------------------------------------------------------------------------------------------ twin agents, SE(3) on, default precision
TWIN = dict(method__voxel_sizes=[16], method__voxel_patch_size=3, method__voxel_patch_stride=4, method__transformer_depth=2,
            method__num_latents=32, rlbench__cameras=['front', 'wrist'], rlbench__camera_resolution=[16, 16],
            method__arm_pred_loss=True, method__crop_target_obj_voxel=True)


def _twin_worker(rank, world, port, out_dir):
    """BASELINE.json configs[2] / [3] control flow on two ranks: the acting (`dominant`) and the stabilizing (`assistive`) agent
    both resident, stepped back to back on this rank's shard, SE(3) augmentation and dropout ON, default bf16x3 precision
    (fp16 conv weight gradients) -- `bench.py --agents 2` at a small geometry."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.pop('VOXACTB_PRECISION', None)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from voxactb_amd import synthetic
    from voxactb_amd.agents.peract_bc import launch_utils as lu
    agents, qas = [], []
    for ai, arm in enumerate(('dominant', 'assistive')):
        cfg = lu.default_cfg(replay__batch_size=B, method__which_arm=arm, **TWIN)
        torch.manual_seed(3000 + 31 * rank + ai)                      # different initialisations: rank 0's are broadcast at build()
        ag = lu.create_agent(cfg)
        ag.build(training=True, device=0)
        agents.append(ag)
        qas.append(ag._pose_agent._qattention_agents[0])
        assert qas[-1]._q.encoder.engine().precision == 'bf16x3' and qas[-1]._transform_augmentation
    torch.manual_seed(500 + rank)                                     # per-rank augmentation / dropout draws
    res = dict(losses=[], w=[], w_before_bad=[], w_after_bad=[], raised=[])
    for step in range(2):
        for ai, ag in enumerate(agents):
            rs = synthetic.make_replay_sample(B, TWIN['rlbench__cameras'], (16, 16), 16, 7, seed=60 + 10 * rank + step + 100 * ai,
                                              arm_pred_loss=True, crop_target_obj_voxel=True, crop_radius=0.3 + 0.1 * ai,
                                              keyframes_near_target=True)
            res['losses'].append(float(ag.update(step, {k: v.to('cuda:0') for k, v in rs.items()})['total_losses']))
    res['w'] = [qa._arena.flat_w.cpu() for qa in qas]
    # rank 1 alone draws a keyframe far outside the scene: its retry budget runs out.  Every rank must skip that optimizer step
    # (the status words are MIN-reduced) and every rank raises at its next update(), as the reference job would die as a whole
    res['w_before_bad'] = qas[0]._arena.flat_w.cpu()
    rs = synthetic.make_replay_sample(B, TWIN['rlbench__cameras'], (16, 16), 16, 7, seed=77 + rank, arm_pred_loss=True,
                                      crop_target_obj_voxel=True, keyframes_near_target=True)
    if rank == 1:
        rs['gripper_pose'][:, :, :3] = -50.0
    agents[0].update(2, {k: v.to('cuda:0') for k, v in rs.items()})
    res['w_after_bad'] = qas[0]._arena.flat_w.cpu()
    try:
        agents[0].update(3, {k: v.to('cuda:0') for k, v in rs.items()})
        res['raised'] = False
    except Exception as e:  # noqa: BLE001
        res['raised'] = 'Failing to perturb' in str(e)
    torch.save(res, os.path.join(out_dir, 'twin%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_twin_agents_with_se3_in_the_default_precision(tmp_path, monkeypatch):
    monkeypatch.delenv('VOXACTB_PRECISION', raising=False)
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_twin_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    r0, r1 = (torch.load(os.path.join(str(tmp_path), 'twin%d.pt' % r)) for r in range(world))
    assert all(np.isfinite(r0['losses'])) and all(np.isfinite(r1['losses']))
    assert r0['losses'] != r1['losses']                                    # different shards, different perturbations
    for a, b in zip(r0['w'], r1['w']):                                     # ... one model per agent on both ranks, bit for bit
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    assert not torch.equal(r0['w'][0], r0['w'][1])                         # the two agents are independent networks
    for r in (r0, r1):                                                     # the failed step touched nothing, on either rank
        assert torch.equal(r['w_before_bad'], r['w_after_bad'])
        assert r['raised'] is True


def test_bench_scaling_command_with_two_ranks_on_one_gpu():
    """The driver's SCALE command at N = 2 -- `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ...` -- with the
    gloo backend (RCCL refuses two ranks per GPU) at the headline size: the launch plumbing, the barrier + max-over-ranks timing, the
    rank-0 JSON line (n_gpus, weak scaling, whole-job value, the gradient-exchange record) cannot fail for the first time on the 8-GPU
    node.  Both ranks' shards are distinct (seeds 100 r + ..)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VOXACTB_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--no-cpu-baseline', '--no-other-modes']
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                    # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['steps'] == 2 and d['warmup'] == 1
    assert np.isfinite(d['value']) and d['value'] > 0 and abs(d['value'] - 2 * 1e3 / d['ms_per_step']) < 1e-6 * d['value'] + 1e-9
    assert d['config']['global_batch'] == 2 * 16 and d['config']['parallelism'] == 'dp2'
    ex = d['gradient_exchange']
    assert ex['world_size_observed'] == 2 and ex['backend'] == 'gloo'
    ag = ex['agents'][0]
    assert ag['steps_timed'] == 2 and len(ag['buckets_bytes']) >= 2 and sum(ag['buckets_bytes'].values()) >= 4 * d['config']['params']
    assert ag['exposed_wait_ms_per_step'] >= 0.0 and set(ag['start_to_done_ms']) == set(ag['buckets_bytes'])
    assert np.isfinite(d['final_loss'])
