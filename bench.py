#!/usr/bin/env python
"""bench.py -- voxel-policy train steps/s at BASELINE.json config 2 (100^3 grid, 4 cams 128x128, B=16 per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = QAttentionPerActBCAgent.update() through the agent stack (PreprocessAgent -> QAttentionStackAgent ->
QAttentionPerActBCAgent): SE(3) augmentation, voxelize, Q-network forward, 6 CE losses, backward, gradient
all-reduce (RCCL, N > 1), fused LAMB, and the runner's `.item()` on the loss (offline_train_runner.py:94) -- on a
synthetic replay batch that is already resident in HBM.  Weak scaling: every rank owns its own B=16 shard
(task_uniform_replay_buffer.py:103-108 semantics), the only exchange is the gradient all-reduce.

Rank 0 prints ONE JSON line (metric/value/... + `roofline` for the dominant kernel + `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector rate
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured achievable)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--voxel-size', type=int, default=100)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--image', type=int, default=128)
    ap.add_argument('--depth', type=int, default=6)
    ap.add_argument('--latents', type=int, default=2048)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-modes', '--no-bf16-mode', dest='no_other_modes', action='store_true',
                    help='skip the secondary measurements in the other precisions and the parity probe')
    ap.add_argument('--cpu-voxel-size', type=int, default=0, help='debug: smaller grid for the CPU baseline leg')
    ap.add_argument('--kernel-table', action='store_true', help='print the per-kernel timing table to stderr')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == a.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % a.gpus
    # one process per GPU; bind before the communicator is created.  (Debug only: VOXACTB_BENCH_BACKEND=gloo lets several
    # ranks share one GPU -- RCCL refuses that -- to exercise the N > 1 control flow on a single-GPU box.)
    backend = os.environ.get('VOXACTB_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world)      # backend "nccl" == RCCL on ROCm

    from voxactb_amd import _lib, synthetic
    from voxactb_amd.agents.peract_bc import launch_utils as lu

    V, B, HW = a.voxel_size, a.batch, a.image
    patch = 5 if V % 5 == 0 else 4
    cfg = lu.default_cfg(method__voxel_sizes=[V], method__voxel_patch_size=5, method__voxel_patch_stride=patch,
                         method__transformer_depth=a.depth, method__num_latents=a.latents, replay__batch_size=B,
                         rlbench__camera_resolution=[HW, HW], ddp__num_devices=world)
    torch.manual_seed(1234)           # identical initial weights on every rank (DDP broadcasts rank 0's upstream)
    agent = lu.create_agent(cfg)
    agent.build(training=True, device=dev_index)
    n_params = sum(p.numel() for p in agent._pose_agent._qattention_agents[0]._q.parameters())
    batches = [{k: v.to(dev) for k, v in synthetic.make_replay_sample(
        B, cfg.rlbench.cameras, (HW, HW), V, 4, seed=100 * rank + j).items()} for j in range(2)]

    eng = agent._pose_agent._qattention_agents[0]._q.encoder.engine()
    headline_mode = eng.precision          # 'bf16x3' unless VOXACTB_PRECISION overrides it
    counter = [0]

    def step():
        i = counter[0]
        counter[0] += 1
        out = agent.update(i, batches[i % 2])
        return float(out['total_losses'])          # the runner's .item() (device sync every step)

    def measure(mode, steps, warmup):
        """W untimed + exactly K timed update() steps in `mode`, bracketed by barrier + synchronize; max over ranks."""
        eng.precision = mode
        for _ in range(warmup):
            step()
        timer = _lib.KernelTimer()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        _lib.TIMER = timer
        t0 = time.perf_counter()
        loss = None
        for _ in range(steps):
            loss = step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        _lib.TIMER = None
        tt = torch.tensor([dt], device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        eng.precision = headline_mode
        return float(tt[0]), loss, timer.summary()

    torch.manual_seed(1000 + rank)    # augmentation draws differ per rank, as they would with per-rank replay shards
    dt, loss, agg = measure(headline_mode, a.steps, a.warmup)

    # secondary measurements of the same workload in the other precisions (never the headline `value`)
    others = {}
    if not a.no_other_modes:
        for mode in ('fp32', 'bf16x3', 'bf16'):
            if mode == headline_mode:
                continue
            dt2, _, agg2 = measure(mode, a.steps, 1)
            others[mode] = {'value': world * a.steps / dt2, 'unit': 'steps/s', 'ms_per_step': dt2 / a.steps * 1e3,
                            'dtype': MODE_DTYPE[mode], 'rooflines': group_rooflines(agg2, mode, a.steps),
                            'note': MODE_NOTE[mode]}

    # how far the headline precision is from the exact-fp32 matrix-core path on THIS workload (forward, eval mode, B=2)
    probe = None
    if rank == 0 and headline_mode != 'fp32' and not a.no_other_modes:
        g = torch.Generator(device='cpu').manual_seed(5)
        grid = (torch.rand(2, V, V, V, 10, generator=g) * (torch.rand(2, V, V, V, 1, generator=g) < 0.05)).to(dev)
        prop = batches[0]['low_dim_state'][:2, 0].float() if batches[0]['low_dim_state'].dim() > 2 else batches[0]['low_dim_state'][:2].float()
        lang = batches[0]['lang_token_embs'][:2, 0].float() if batches[0]['lang_token_embs'].dim() > 3 else batches[0]['lang_token_embs'][:2].float()
        qs = {}
        for mode in (headline_mode, 'fp32'):
            eng.precision = mode
            outs, _ = eng.forward(grid, prop, lang, training=False, save=False)
            qs[mode] = [o.float().clone() for o in outs[:3]]
        eng.precision = headline_mode
        probe = {'what': 'max |Q(%s) - Q(exact fp32 MFMA)| over q_trans / rot_grip / collision, V=%d forward, B=2, eval' % (headline_mode, V),
                 'q_trans': float((qs[headline_mode][0] - qs['fp32'][0]).abs().max()),
                 'rot_grip': float((qs[headline_mode][1] - qs['fp32'][1]).abs().max()),
                 'collision': float((qs[headline_mode][2] - qs['fp32'][2]).abs().max()),
                 'q_trans_abs_max': float(qs['fp32'][0].abs().max()), 'bound': 1e-4}

    # act() latency: eval agent, B=1 observation with precomputed language embeddings (the CLIP text
    # encoder's weights are not in either tree), 1 voxelize + 1 forward + argmax + the D2H copy of the 9-vector action
    act_lat = None
    if rank == 0 and not a.no_other_modes:
        ev = lu.create_agent(cfg)
        ev.build(training=False, device=dev_index)
        rs = synthetic.make_replay_sample(1, cfg.rlbench.cameras, (HW, HW), V, 4, seed=3)
        obs = {k: v.to(dev) for k, v in rs.items() if k.endswith(('_rgb', '_point_cloud')) or k == 'low_dim_state'}
        obs = {k: v.unsqueeze(0) if v.dim() < 5 and k != 'low_dim_state' else v for k, v in obs.items()}
        obs['low_dim_state'] = rs['low_dim_state'].to(dev)
        obs['lang_goal_emb'] = rs['lang_goal_emb'][0].to(dev)
        obs['lang_token_embs'] = rs['lang_token_embs'][0].to(dev)
        for i in range(3):
            ev.act(i, dict(obs), deterministic=True)      # act() annotates the dict it is given
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            ev.act(i, dict(obs), deterministic=True)      # act() annotates the dict it is given
        torch.cuda.synchronize()
        act_lat = {'ms_per_act': (time.perf_counter() - t0) / 10 * 1e3, 'precision': ev._pose_agent._qattention_agents[0]._q.encoder.engine().precision,
                   'what': 'QAttentionPerActBCAgent.act(): B=1, V=%d, %d cams %dx%d, language embeddings given' % (V, len(cfg.rlbench.cameras), HW, HW)}
        del ev

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = world * a.steps / dt
        tot_ms = sum(d['ms'] for d in agg.values())
        roofs = group_rooflines(agg, headline_mode, a.steps)
        dom_key = max(roofs, key=lambda k: roofs[k]['ms_per_step'])
        roofline = dict(roofs[dom_key])
        roofline['kernel'] = dom_key
        roofline['traffic'] = None
        if headline_mode == 'bf16x3' and V == 100 and B == 16 and dom_key.startswith('conv3d'):
            # HBM bytes per launch of the dominant kernel of this group (conv3_halo_kernel<2,1,4,1,0>: final fwd and final
            # dgrad + fused padding adjoint), from the committed PMC passes of this same command (profiles/
            # r01_pmc_*_v8.txt; separate --pmc FETCH_SIZE / WRITE_SIZE runs, KiB units).  FETCH_SIZE doubled as
            # MI355X_MICROARCH.md prescribes for 16-B/lane reads on gfx950; WRITE_SIZE is exact (4.0 GiB forward,
            # 2 x 4.0 GiB for the two gradients of the fused dgrad -> mean 6.0M KiB).
            roofline['traffic'] = (2 * 8605105.3 + 6000000.0) * 1024.0
            roofline['traffic_note'] = ('conv3_halo_kernel<2,1,4,1,0> mean per launch: FETCH_SIZE 8.81 GB raw (x2 = 17.6 GB) + '
                                        'WRITE_SIZE 6.14 GB (exactly the outputs); compulsory reads are 8.2 GB (fwd: two 64-channel '
                                        'sources) and 4.1 + 8.2 GB (dgrad: dY once per 64-column block + the two accumulate / mask '
                                        'operands); 23.8 GB / 19.9 ms = 1.2 TB/s, i.e. the kernel is matrix-core-bound, not HBM-bound')
        roofline['share_of_device_time'] = roofline['ms_per_step'] * a.steps / tot_ms
        extra = {k: v for k, v in roofs.items() if k != dom_key}
        if 'voxelize' in agg:
            v = agg['voxelize']
            gbps = v['bytes'] / (v['ms'] * 1e-3) / 1e9
            extra['voxel_scatter'] = {'bound': 'hbm', 'achieved': gbps, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                                      'frac': gbps / PEAK_HBM_GBPS, 'avg_launch_ms': v['ms'] / v['calls'], 'traffic': None}
        if a.kernel_table:
            for label, d in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
                sys.stderr.write('%-44s calls %5d  %9.2f ms/step  %7.2f TF/s\n' % (
                    label, d['calls'] // a.steps, d['ms'] / a.steps, d['flops'] / max(d['ms'], 1e-9) / 1e9))
        cpu = None
        if not a.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N = 1 only
            cpu = cpu_baseline(a, cfg)
        out = {
            'metric': 'voxel-policy train steps/sec (100^3 grid, 4 cams, B=16)', 'value': value, 'unit': 'steps/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': MODE_DTYPE[headline_mode], 'data': 'synthetic',
            'config': {'workload': ('BASELINE.json configs[1]' if (V, B, HW) == (100, 16, 128) else
                                    'BASELINE.json configs[4] shape (per GPU)' if (V, B) == (200, 8) else 'custom size') +
                                   ': QAttentionPerActBCAgent.update(), V=%d, %d cams %dx%d, '
                                   'B=%d per GPU, PerceiverIO depth %d, %d latents, SE(3) aug + dropout on, LAMB'
                                   % (V, len(cfg.rlbench.cameras), HW, HW, B, a.depth, a.latents),
                       'global_batch': B * world, 'parallelism': 'dp%d' % world, 'params': n_params,
                       'precision': headline_mode},
            'samples_per_s': value * B, 'final_loss': loss, 'device_time_ms_per_step': tot_ms / a.steps,
            'roofline': roofline, 'rooflines_other': extra, 'cpu_baseline': cpu, 'precision_note': MODE_NOTE[headline_mode],
            'parity_probe': probe, 'act_latency': act_lat, 'other_precisions': others,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()                   # rank 0 did the rank-0-only extras; leave together
        dist.destroy_process_group()


MODE_DTYPE = {
    'fp32': 'f32 (v_mfma_f32_32x32x2_f32 everywhere)',
    'bf16x3': 'f32 storage / accumulate; matrix products as bf16x3 split (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16)',
    'bf16': 'bf16 matrix cores (fp32 accumulate) for convs, large linears and fused attention; everything else f32',
}
MODE_NOTE = {
    'fp32': 'exact fp32 matrix cores: the reference-parity mode of the first measurements (tests/test_encoder_gpu.py, 1e-4)',
    'bf16x3': 'held to the same 1e-4 Q-value / 2e-5 per-op bounds as the exact-fp32 mode (tests/test_encoder_gpu.py::'
              'test_encoder_fixtures_bf16x3_split_mode, test_bf16x3_gpu.py, test_flash_x3_gpu.py); 3 MFMAs per product',
    'bf16': 'throughput mode, NOT held to the 1e-4 Q-value bound (tests/test_bf16_mode_gpu.py: ~4e-3 on q_trans)',
}
# matrix-core roof per algorithmic FLOP: fp32 MFMA; bf16 dense MFMA / 3 instructions per product; bf16 dense MFMA
MODE_PEAK = {'fp32': PEAK_FP32_MFMA_TFLOPS, 'bf16x3': PEAK_BF16_MFMA_TFLOPS / 3.0, 'bf16': PEAK_BF16_MFMA_TFLOPS}


def group_rooflines(agg, mode, steps):
    """conv / linear / attention kernel groups: algorithmic TFLOP/s against the matrix-core roof of the precision."""
    out = {}
    for key, pref in (('conv3d (implicit-GEMM + LDS-halo launches)', 'conv3'), ('gemm (linear layers)', 'gemm'),
                      ('attention (QK^T, PV and their gradients)', 'attn_core')):
        ms = sum(d['ms'] for l, d in agg.items() if l.startswith(pref))
        fl = sum(d['flops'] for l, d in agg.items() if l.startswith(pref))
        calls = sum(d['calls'] for l, d in agg.items() if l.startswith(pref))
        if ms > 0:
            peak = MODE_PEAK[mode]
            tf = fl / (ms * 1e-3) / 1e12
            out[key] = {'bound': 'mfma', 'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tf / peak,
                        'peak_basis': {'fp32': 'fp32 MFMA 157.3', 'bf16x3': 'bf16 dense MFMA 2500 / 3 MFMAs per product',
                                       'bf16': 'bf16 dense MFMA 2500'}[mode],
                        'launches': calls // steps, 'avg_launch_ms': ms / max(calls, 1), 'ms_per_step': ms / steps}
    return out


def cpu_baseline(a, cfg):
    """The oracle (CPU restatement of the reference path, kind 'port') timed on this host: ONE sample (B=1) of the same
    workload through voxelize + forward + losses + backward + LAMB; a B=16 step is 16x that."""
    import torch
    from oracle import agent as oagent, perceiver as operc, weights as ow
    from voxactb_amd import synthetic
    # measured on the GPU box host (256 logical CPUs): PyTorch CPU fwd at B=1 takes 3.0 / 2.6 / 3.6 / 7.1 s with
    # 16 / 32 / 64 / 128 threads -- oversubscription hurts, so the baseline uses the best setting, 32 threads
    ncores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(ncores)
    V = a.cpu_voxel_size or a.voxel_size
    s = 5 if V % 5 == 0 else 4
    shapes = operc.param_shapes(a.depth, V, 4, num_latents=a.latents, voxel_patch_size=5, voxel_patch_stride=s)
    P = ow.hashed_state_dict(shapes, 0)
    rs = synthetic.make_replay_sample(1, cfg.rlbench.cameras, (a.image, a.image), V, 4, seed=7)
    r = {k: (v[:, 0] if v.dim() > 2 else v) for k, v in rs.items()}
    r = {k: ((v.float() / 255.0) * 2.0 - 1.0 if 'rgb' in k else v.float()) for k, v in r.items()}
    bt = dict(pcd=[r['%s_point_cloud' % c] for c in cfg.rlbench.cameras], rgb=[r['%s_rgb' % c] for c in cfg.rlbench.cameras],
              proprio=r['low_dim_state'], lang_token_embs=r['lang_token_embs'], bounds=torch.tensor([synthetic.SCENE_BOUNDS]),
              trans=r['trans_action_indicies'], rot_grip=r['rot_grip_action_indicies'], ignore_collisions=r['ignore_collisions'])
    t0 = time.perf_counter()
    oagent.train_steps(P, [bt], V, 1, depth=a.depth, voxel_patch_stride=s)
    dt = time.perf_counter() - t0
    return {'value': 1.0 / (dt * a.batch), 'unit': 'steps/s', 'cores': ncores, 'kind': 'port',
            'sample': 'one replay sample (B=1) of the same workload through the CPU oracle (voxelize + fwd + 6 CE + bwd + LAMB) '
                      'took %.1f s; a B=%d step is %dx that' % (dt, a.batch, a.batch), 'seconds_per_sample': dt}


if __name__ == '__main__':
    main()
