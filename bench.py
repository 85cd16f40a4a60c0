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

    python bench.py --agents 2 --aug-copies 4        # BASELINE.json configs[2] / [3]: the acting + stabilizing twin agents
(low_dim 7, arm-prediction loss, per-sample crop bounds), both resident on every GPU; one step = each agent updating on 4
independently SE(3)-perturbed copies of its 16 replay samples (4 update() calls per agent, 128 voxel grids per step).

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
    ap.add_argument('--cams', type=int, default=4, choices=(3, 4), help='3 = front | wrist | wrist2: the cameras of the released VoxAct-B recipe '
                    '(scripts/train_open_jar_ours_vlm_10_demos_v2_11_acting.sh:12; with --voxel-size 50 --batch 1 --release-recipe)')
    ap.add_argument('--release-recipe', action='store_true', help='the single released agent: which_arm dominant, 7-dim proprioception, arm loss, '
                    'crop bounds, aug_rpy [0, 0, 45] (same script, lines 18-27)')
    ap.add_argument('--agents', type=int, default=1, choices=(1, 2),
                    help='2 = the acting + stabilizing twin agents of BASELINE.json configs[2] / [3]')
    ap.add_argument('--aug-copies', type=int, default=1,
                    help='SE(3)-perturbed copies of every replay sample per step (update() calls per agent and step)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-modes', '--no-bf16-mode', dest='no_other_modes', action='store_true',
                    help='skip the secondary measurements in the other precisions and the parity probe')
    ap.add_argument('--cpu-voxel-size', type=int, default=0, help='debug: smaller grid for the CPU baseline leg')
    ap.add_argument('--cpu-full-batch', action='store_true', help='cpu_baseline: also time ONE real B=16 step of the oracle (needs >= 64 GB '
                    'of host RAM and minutes; SURVEY.md 8d "if RAM allows")')
    ap.add_argument('--kernel-table', action='store_true', help='print the per-kernel timing table to stderr')
    ap.add_argument('--replay-stream', action='store_true',
                    help='feed the HEADLINE region from the replay store: a fresh task-uniform batch per step through '
                         'ShardReplayBuffer + DeviceBatchStream (host gather into pinned memory, H2D on a side stream) instead of '
                         'two batches resident in HBM.  Without the flag the same measurement is reported beside the headline '
                         'as `replay_stream` (offline_train_runner.py:136-143: Sample time + Step time)')
    ap.add_argument('--replay-rows', type=int, default=96, help='distinct synthetic transitions in the replay store of --replay-stream')
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

    from voxactb_amd import _lib, ops, synthetic
    from voxactb_amd.agents.peract_bc import launch_utils as lu

    V, B, HW = a.voxel_size, a.batch, a.image
    patch = 5 if V % 5 == 0 else 4
    twin = a.agents == 2
    over = dict(method__voxel_sizes=[V], method__voxel_patch_size=5, method__voxel_patch_stride=patch,
                method__transformer_depth=a.depth, method__num_latents=a.latents, replay__batch_size=B,
                rlbench__camera_resolution=[HW, HW], ddp__num_devices=world)
    if a.cams == 3:
        over['rlbench__cameras'] = ['front', 'wrist', 'wrist2']
    twin = twin or a.release_recipe          # (the released agent is one of the twins)
    # twin agents: `which_arm` dominant (acting) / assistive (stabilizing), 7-dim proprioception, arm-prediction head,
    # grid cropped around the target object (scripts/train_open_jar_ours_vlm_10_demos_v2_11_{acting,stabilizing}.sh:22-24)
    arms = ['dominant'] if a.release_recipe else ['dominant', 'assistive'] if twin else ['right']
    agents, cfgs = [], []
    torch.manual_seed(1234)           # (the agent broadcasts rank 0's weights at build(), as DDP does upstream)
    for arm in arms:
        cfg = lu.default_cfg(**over) if not twin else lu.default_cfg(
            method__which_arm=arm, method__arm_pred_loss=True, method__crop_target_obj_voxel=True, **over)
        if a.release_recipe:
            cfg.method.transform_augmentation.aug_rpy = [0.0, 0.0, 45.0]
        ag = lu.create_agent(cfg)
        ag.build(training=True, device=dev_index)
        agents.append(ag)
        cfgs.append(cfg)
    cfg, agent = cfgs[0], agents[0]
    low_dim = 7 if twin else 4
    n_params = sum(p.numel() for p in agent._pose_agent._qattention_agents[0]._q.parameters())
    # open_jar-like / open_drawer-like scene bounds for the two halves of the ranks
    scene = synthetic.SCENE_BOUNDS if (not twin or rank % 2 == 0) else [-0.8, -1.0, 0.1, 1.2, 1.0, 2.8]
    batches = [[{k: v.to(dev) for k, v in synthetic.make_replay_sample(
        B, cfg.rlbench.cameras, (HW, HW), V, low_dim, seed=100 * rank + 10 * ai + j, scene_bounds=scene, arm_pred_loss=twin,
        crop_target_obj_voxel=twin, crop_radius=0.3 if ai == 0 else 0.4, keyframes_near_target=twin).items()} for j in range(2)] for ai in range(len(agents))]

    arenas = [ag._pose_agent._qattention_agents[0]._arena for ag in agents]
    engines = [ag._pose_agent._qattention_agents[0]._q.encoder.engine() for ag in agents]
    eng = engines[0]
    headline_mode = eng.precision          # 'bf16x3' unless VOXACTB_PRECISION overrides it
    headline_bwd = eng.bwd_precision       # '' (= same as the forward) unless VOXACTB_BWD_PRECISION overrides it
    headline_attn = eng.attn_kernel        # 'auto' (round 6: the pipelined single-fp16 attention forward from 2^22 scores on) unless VOXACTB_ATTN_KERNEL overrides it
    counter = [0]
    updates_per_step = len(agents) * a.aug_copies

    streams = [None]                  # [list of one DeviceBatchStream per agent] while a replay-stream region is measured
    sample_s = [0.0]

    def step():
        i = counter[0]
        counter[0] += 1
        loss = 0.0
        for ai, ag in enumerate(agents):           # the twin agents are stepped back to back on this GPU's shard
            if streams[0] is not None:
                t_s = time.perf_counter()
                batch = next(streams[0][ai])       # offline_train_runner.py:137-140: next(data_iter) + .to(device) (already there)
                sample_s[0] += time.perf_counter() - t_s
            else:
                batch = batches[ai][i % 2]
            for _ in range(a.aug_copies):          # every call draws a fresh SE(3) perturbation of the same replay samples
                out = ag.update(i, dict(batch))
                loss = float(out['total_losses'])  # the runner's .item() (device sync every update)
        return loss

    def open_streams():
        """one replay store per agent, filled with --replay-rows synthetic transitions of the configs' schema, task-uniform over
        two tasks, rank-strided (task_uniform_replay_buffer.py:103-108); batches come out of DeviceBatchStream on the device"""
        import numpy as np
        from voxactb_amd import replay as R
        out = []
        for ai, (ag, cf) in enumerate(zip(agents, cfgs)):
            buf = lu.create_replay(B, 1, False, True, None, cf.rlbench.cameras, [V], [HW, HW], which_arm=cf.method.which_arm,
                                   crop_target_obj_voxel=cf.method.crop_target_obj_voxel, arm_pred_loss=cf.method.arm_pred_loss,
                                   arm_id_to_proprio=cf.method.arm_id_to_proprio)
            buf._rank, buf._num_replicas = rank, world
            n = a.replay_rows
            src = synthetic.make_replay_sample(n, cf.rlbench.cameras, (HW, HW), V, low_dim, seed=4242 + ai, scene_bounds=scene,
                                               arm_pred_loss=twin, crop_target_obj_voxel=twin, crop_radius=0.3 if ai == 0 else 0.4,
                                               keyframes_near_target=twin)
            names = [e.name for e in buf._observation_elements]
            for i in range(n):
                row = {}
                for nm in names:
                    if nm == 'task':
                        row[nm] = 'open_jar' if i % 2 else 'open_drawer'
                    elif nm == 'lang_goal':
                        row[nm] = np.array(['open it'], dtype=object)
                    else:
                        row[nm] = src[nm][i, 0].numpy()
                if i % 6 == 5:
                    buf.add_final(**row)
                else:
                    buf.add(np.zeros(8, np.float32), 0.0, i % 6 == 4, False, demo=True, **row)
            buf.seed(977 + 13 * rank + ai)
            out.append(R.DeviceBatchStream(buf, device=dev_index, depth=2))
        return out

    def measure(mode, steps, warmup, only=None, stream=False):
        """W untimed + exactly K timed steps in `mode`, bracketed by barrier + synchronize; max over ranks.  `only`: the timer
        labels whose launches are bracketed by HIP events (None = every launch).  stream: batches from the replay store."""
        base_mode, _, attn = mode.partition('+')                         # 'bf16x3+attn_x3' = round 3's bf16x3 attention forward (VOXACTB_ATTN_KERNEL=r3)
        for e_ in engines:
            e_.precision, _, e_.bwd_precision = base_mode.partition('/')  # 'bf16x3/bf16' = forward bf16x3, backward products bf16
            e_.attn_kernel = 'auto' if attn == 'attn_f16' else 'r3' if attn == 'attn_x3' else headline_attn
        wino = (ops.FINAL_WINOGRAD, ops.DGRAD_WINOGRAD)
        if attn == 'direct_final':                                       # same-run A/B: `final` forward + d(u0) on the direct LDS-halo kernels
            ops.FINAL_WINOGRAD = ops.DGRAD_WINOGRAD = False
        if stream:
            streams[0] = open_streams()
        for _ in range(warmup):
            step()
        sample_s[0] = 0.0
        timer = _lib.KernelTimer(only)
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
        for e_ in engines:
            e_.precision, e_.bwd_precision, e_.attn_kernel = headline_mode, headline_bwd, headline_attn
        ops.FINAL_WINOGRAD, ops.DGRAD_WINOGRAD = wino
        if stream:
            for st_ in streams[0]:
                st_.close()
            streams[0] = None
        return float(tt[0]), loss, timer.summary()

    torch.manual_seed(1000 + rank)    # augmentation draws differ per rank, as they would with per-rank replay shards
    # (1) profile pass, untimed for the headline: EVERY launch between two HIP events (~1100 launches per step; the event pairs
    #     cost several ms per step) -> per-kernel table, kernel groups, and which label is the dominant kernel
    prof_steps = min(a.steps, 4)
    dt_prof, _, agg = measure(headline_mode, prof_steps, a.warmup)
    dom_label = max((l for l in agg if agg[l]['flops'] > 0), key=lambda l: agg[l]['ms'])
    # (2) THE timed region: exactly K steps; only the dominant kernel and the voxelizer are event-timed inside it
    for ar in arenas:
        ar.timing = world > 1         # N > 1: events around the bucketed gradient exchange of the timed region (flat_params.py)
    dt, loss, agg_head = measure(headline_mode, a.steps, 1 if a.replay_stream else 0, only={dom_label, 'voxelize'}, stream=a.replay_stream)
    exchange = None
    if world > 1:
        exchange = {'backend': dist.get_backend(), 'world_size_observed': dist.get_world_size(), 'rank': rank,
                    'agents': [ar.exchange_summary() for ar in arenas]}
        for ar in arenas:
            ar.timing = False
    sample_ms_head = sample_s[0] / a.steps * 1e3
    # (3) the same K steps fed from the replay store (fresh batch per step, host gather + H2D inside the region) -- reported
    #     beside the headline, never as `value` (inputs resident in HBM is the contract of `value`)
    replay_stream = None
    if not a.no_other_modes and not a.replay_stream:
        dt_rs, _, _ = measure(headline_mode, a.steps, 1, only=set(), stream=True)
        replay_stream = {'value': world * a.steps / dt_rs, 'unit': 'steps/s', 'ms_per_step': dt_rs / a.steps * 1e3,
                         'sample_ms_per_step': sample_s[0] / a.steps * 1e3, 'rows_in_store': a.replay_rows,
                         'what': 'the same K steps with a fresh task-uniform batch per step from ShardReplayBuffer through '
                                 'DeviceBatchStream (host gather into pinned staging, H2D on a side stream, rank-strided rows): '
                                 'sample_ms_per_step = host time blocked in next(data_iter), the reference\'s "Sample time" '
                                 '(offline_train_runner.py:136-143)'}

    # secondary measurements of the same workload in the other precisions (never the headline `value`)
    others = {}
    if not a.no_other_modes:
        for mode in ('fp32', 'bf16x3', 'bf16x3+attn_f16', 'bf16x3+attn_x3', 'bf16x3+direct_final', 'bf16x3/bf16', 'bf16'):
            if mode == headline_mode and not headline_bwd:
                continue
            if mode == 'bf16x3+attn_f16' and (headline_mode != 'bf16x3' or headline_attn != 'r3'):
                continue
            if mode == 'bf16x3+attn_x3' and (headline_mode != 'bf16x3' or headline_attn != 'auto'):
                continue
            if mode == 'bf16x3+direct_final' and (headline_mode != 'bf16x3' or not (ops.FINAL_WINOGRAD or ops.DGRAD_WINOGRAD) or V % 2):
                continue
            dt2, _, agg2 = measure(mode, a.steps, 1)
            others[mode] = {'value': world * a.steps / dt2, 'unit': 'steps/s', 'ms_per_step': dt2 / a.steps * 1e3,
                            'dtype': MODE_DTYPE[mode], 'rooflines': group_rooflines(agg2, mode.partition('/')[0].partition('+')[0], a.steps),
                            'note': MODE_NOTE[mode]}

    # parity of the headline precision against the REFERENCE at this geometry: the digest the reference produced for a
    # seeded batch with name-hashed weights (tests/golden/f5_encoder_c2_digest.npz, generated by make_golden.py from the
    # reference's own modules; the same check runs as tests/test_c2_reference_gpu.py in both precisions)
    probe = None
    if rank == 0 and not a.no_other_modes:
        # the timed region's linear layers run B * latents rows: from ops.WIDE_MIN_M rows on that is the 128 x 512-tile dispatch -- the B = 1
        # digest is therefore forced through the same kernels, and the B = 8 reference fixture (16 384 rows) takes them by itself
        from voxactb_amd import ops
        wide = B * a.latents >= ops.WIDE_MIN_M
        probe = reference_digest_check(dev, headline_mode, V, a.depth, a.latents, HW, 'f5_encoder_c2_digest', force_wide=wide)
        if probe is not None:
            probe['b8'] = reference_digest_check(dev, headline_mode, V, a.depth, a.latents, HW, 'f5gb8_encoder_c2_b8_grads')

    # act() latency: eval agent, B=1 observation, 1 voxelize + 1 forward + argmax + the D2H copy of the 9-vector
    # action -- with precomputed language embeddings, and with the CLIP text transformer in front (helpers/clip_text.py; the RN50
    # checkpoint is in neither tree, so its text half runs on name-hashed weights of the same shapes)
    act_lat = None
    if rank == 0 and not a.no_other_modes:
        ev = lu.create_agent(cfg)
        ev.build(training=False, device=dev_index)
        rs = synthetic.make_replay_sample(1, cfg.rlbench.cameras, (HW, HW), V, 4, seed=3)
        obs = {k: v.to(dev) for k, v in rs.items()
               if k.endswith(('_rgb', '_point_cloud', '_camera_extrinsics', '_camera_intrinsics')) or k == 'low_dim_state'}
        for cam in cfg.rlbench.cameras:
            obs['%s_camera_extrinsics' % cam][0, 0, 2, 3] = -1.0             # (a camera 1 m off the scene origin)
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
        from voxactb_amd.helpers.clip_text import ClipTextEncoder
        text = ClipTextEncoder(synthetic.hashed_clip_text_state_dict(), dev)
        ev._pose_agent._qattention_agents[0].set_text_encoder(text.for_agent())
        obs_t = {k: v for k, v in obs.items() if k not in ('lang_goal_emb', 'lang_token_embs')}
        tok = torch.zeros((1, 1, 77), dtype=torch.long, device=dev)
        tok[0, 0, :5] = torch.tensor([49406, 1000, 2000, 3000, 49407])
        obs_t['lang_goal_tokens'] = tok
        for i in range(3):
            ev.act(i, dict(obs_t), deterministic=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            ev.act(i, dict(obs_t), deterministic=True)
        torch.cuda.synchronize()
        act_lat['ms_per_act_with_text_encoder'] = (time.perf_counter() - t0) / 10 * 1e3      # unchanged instruction: cached encoding
        uncached = text.for_agent(cache=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            uncached(tok[0, 0])
        torch.cuda.synchronize()
        act_lat['ms_text_encode'] = (time.perf_counter() - t0) / 10 * 1e3
        del ev, text

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = world * a.steps / dt
        tot_ms = sum(d['ms'] for d in agg.values())
        roofs = group_rooflines(agg, headline_mode, prof_steps)
        # the dominant KERNEL = the timer label (one C-ABI entry point at one shape) with the largest share of device time;
        # the conv / linear / attention GROUP figures stay in `rooflines_other`
        d = agg_head[dom_label]
        tf = d['flops'] / (d['ms'] * 1e-3) / 1e12
        roofline = {'bound': 'mfma', 'achieved': tf, 'peak': MODE_PEAK[headline_mode], 'unit': 'TFLOP/s',
                    'frac': tf / MODE_PEAK[headline_mode], 'peak_basis': MODE_PEAK_BASIS[headline_mode],
                    'kernel': dom_label, 'entry': d['entry'], 'launches': d['calls'] // a.steps,
                    'avg_launch_ms': d['ms'] / max(d['calls'], 1), 'ms_per_step': d['ms'] / a.steps,
                    'algorithmic_flops_per_launch': d['flops'] / max(d['calls'], 1), 'traffic': None}
        dom_key = dom_label
        if headline_mode == 'bf16x3' and (V, B) == (100, 16):
            tr = profile_traffic(dom_label)
            if tr is not None:
                roofline['traffic'], roofline['traffic_note'] = tr
        if headline_mode == 'bf16x3':
            roofline['frac_of_x3_roof'] = tf / (PEAK_BF16_MFMA_TFLOPS / 3.0)
        from voxactb_amd import ops as _ops
        if dom_label.startswith('conv3d_bf16[k3 s1 128->64') and _ops.FINAL_WINOGRAD and V % 2 == 0:
            roofline['arithmetic_note'] = ('achieved / frac count the DIRECT convolution\'s flops (2 x 27 x Cin x Cout per output voxel); the kernel '
                                           'evaluates the depth taps by Winograd F(2, 3) and issues 2/3 of them as MFMA work (x 3 for bf16x3)')
            # (round-5 advisor) the MFMA work the kernel EXECUTES next to the algorithmic figure: 36 x 2 instead of 27 x 4 MFMA groups per
            # wave and chunk, three bf16 MFMAs per product
            roofline['executed_mfma_flops_per_launch'] = roofline['algorithmic_flops_per_launch'] * (72.0 / 108.0) * (3.0 if headline_mode == 'bf16x3' else 1.0)
            roofline['executed_mfma_frac_of_peak'] = roofline['executed_mfma_flops_per_launch'] / (roofline['avg_launch_ms'] * 1e-3) / 1e12 / MODE_PEAK[headline_mode]
        roofline['share_of_device_time'] = agg[dom_label]['ms'] / tot_ms
        extra = dict(roofs)
        # BASELINE.json north_star's three numeric targets, nested in `roofline` so the driver's parsed record keeps them (round-5 review):
        # attention >= 0.30 of the dense bf16 MFMA peak, voxel scatter >= 0.60 of the HBM peak, >= 6 x at 8 GPUs (not measurable on one GPU)
        north = {'targets': {'attention_frac_of_bf16_mfma_peak': 0.30, 'voxel_scatter_frac_of_hbm_peak': 0.60, 'scaling_at_8_gpus': 6.0}}
        att = roofs.get('attention (QK^T, PV and their gradients)')
        if att is not None:
            north['attention_in_step'] = {'frac': att['frac'], 'achieved_tflops': att['achieved'], 'ms_per_step': att['ms_per_step'],
                                          'launches': att['launches'],
                                          'basis': 'algorithmic flops of QK^T, PV and their gradients (4 / 10 Nq Nk d per batch and head) over the '
                                                   'summed durations of the attention-core launches of the event-timed pass of THIS run, / 2500 TF/s'}
        if 'voxelize' in agg_head:
            v = agg_head['voxelize']
            extra['voxel_scatter'] = voxel_roofline(v['ms'] / v['calls'], v['bytes'] / v['calls'], V, B, True)
            vs = extra['voxel_scatter']
            north['voxel_scatter_in_step'] = {'avg_launch_ms': vs['avg_launch_ms'], 'algorithmic_equivalent_frac': vs['algorithmic_equivalent_gbps'] / PEAK_HBM_GBPS,
                                              'real_traffic_frac': vs.get('frac'),
                                              'basis': 'the training path\'s call (persistent grids updated in place), timed INSIDE the K-step region; '
                                                       'algorithmic_equivalent_frac = (read N*6*4 + write V^3*10*4 bytes per sample) / time / 8 TB/s -- the call '
                                                       'moves fewer bytes than that (real_traffic_frac, PMC)'}
            if not a.no_other_modes:
                extra['voxel_scatter_stateless'] = voxel_stateless(agents[0], batches[0][0], cfg, V, B)
                north['voxel_scatter_stateless'] = {'avg_launch_ms': extra['voxel_scatter_stateless']['avg_launch_ms'],
                                                    'frac': extra['voxel_scatter_stateless']['frac'],
                                                    'basis': 'fresh grid, every cell written (coords_to_bounding_voxel_grid / act()): algorithmic bytes / time / 8 TB/s'}
                extra['attention_kernels_B16_H8_N2048_d64'] = attention_kernel_probe(dev)
                north['attention_kernels_alone'] = {k.split(' ')[0]: round(x['frac'], 4) for k, x in extra['attention_kernels_B16_H8_N2048_d64'].items()}
        north['scaling_at_8_gpus'] = None if world == 1 else {'n_gpus': world, 'note': 'the driver computes efficiency from the per-N values'}
        roofline['north_star'] = north
        if a.kernel_table:
            for label, d in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
                sys.stderr.write('%-44s calls %5d  %9.2f ms/step  %7.2f TF/s\n' % (
                    label, d['calls'] // prof_steps, d['ms'] / prof_steps, d['flops'] / max(d['ms'], 1e-9) / 1e9))
        cpu = None
        if not a.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N = 1 only
            cpu = cpu_baseline(a, cfg)
        out = {
            'metric': 'voxel-policy train steps/sec (100^3 grid, 4 cams, B=16)', 'value': value, 'unit': 'steps/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': MODE_DTYPE[headline_mode], 'data': 'synthetic',
            'config': {'workload': ('released VoxAct-B recipe (train_open_jar_ours_vlm_10_demos_v2_11_acting.sh: one dominant-arm agent, V=50, '
                                    'front | wrist | wrist2, replay batch 1, low_dim 7, arm loss, crop bounds, aug_rpy [0, 0, 45]); ' if a.release_recipe else
                                    'BASELINE.json configs[2] / [3] (twin acting + stabilizing agents, low_dim 7, arm loss, crop bounds; '
                                    'one step = %d agents x %d SE(3)-perturbed copies = %d update() calls of B=%d; ' % (
                                        len(agents), a.aug_copies, updates_per_step, B) if twin else '') +
                                   ('BASELINE.json configs[1]' if (V, B, HW) == (100, 16, 128) else
                                    'BASELINE.json configs[4] shape (per GPU)' if (V, B) == (200, 8) else 'custom size') +
                                   ': QAttentionPerActBCAgent.update(), V=%d, %d cams %dx%d, '
                                   'B=%d per GPU, PerceiverIO depth %d, %d latents, SE(3) aug + dropout on, LAMB'
                                   % (V, len(cfg.rlbench.cameras), HW, HW, B, a.depth, a.latents),
                       'global_batch': B * world, 'parallelism': 'dp%d' % world, 'params': n_params,
                       'precision': headline_mode},
            'samples_per_s': value * B * updates_per_step, 'updates_per_step': updates_per_step, 'final_loss': loss,
            'device_time_ms_per_step': tot_ms / prof_steps, 'ms_per_step_profile_pass': dt_prof / prof_steps * 1e3,
            'timing_note': 'value / ms_per_step / roofline / voxel_scatter: the K-step timed region, in which only the dominant kernel and '
                           'the voxelizer are bracketed by HIP events; device_time_ms_per_step, rooflines_other and the kernel table: a '
                           'separate %d-step pass with every launch event-timed (ms_per_step_profile_pass: ~2 200 event records per step make that pass '
                           'host-bound, it is not a throughput figure)' % prof_steps,
            'input': ('replay-stream: fresh batch per step from ShardReplayBuffer via DeviceBatchStream (H2D inside the timed region), '
                      'sample_ms_per_step %.3f' % sample_ms_head) if a.replay_stream else 'two synthetic batches resident in HBM, alternated',
            'replay_stream': replay_stream, 'gradient_exchange': exchange,
            'shard_note': None if world == 1 else 'rank r steps replay batches seeded 100 r + ..: %d distinct shards, no data-path collective except '
                                                  'the per-bucket gradient all-reduce' % world,
            'roofline': roofline, 'rooflines_other': extra, 'cpu_baseline': cpu, 'precision_note': MODE_NOTE[headline_mode],
            'parity_vs_reference': probe, 'act_latency': act_lat, 'other_precisions': others,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()                   # rank 0 did the rank-0-only extras; leave together
        dist.destroy_process_group()


# HBM bytes per launch of the dominant kernel: read from the NEWEST committed PMC passes of this command at configs[1]
# (profiles/rNN_vK_pmc_{FETCH,WRITE}_SIZE_summary.txt: separate rocprofv3 --pmc runs, tools/profile_round.sh; KiB per dispatch; FETCH_SIZE
# times the factor of TRAFFIC_KERNELS below -- MI355X_MICROARCH.md's x 2 for wide streaming reads on gfx950 unless calibrated otherwise --,
# WRITE_SIZE as reported) -- timer label -> the kernels one call
# of that C-ABI entry launches.  tests/test_bench_traffic_cpu.py checks that every mapped kernel is present in the newest profile.
TRAFFIC_KERNELS = {
    # label prefix -> (kernels of one call, what the compulsory traffic is, FETCH_SIZE factor).  Factor 2 = the guide's rule for wide
    # coalesced streaming reads (128-byte requests tallied at 64 bytes); factor 1 for the LDS-halo conv kernels, whose loads are 64-byte
    # segments (16 channels of a voxel row per chunk): calibrated as the guide asks for other access widths -- measured in round 6 on a known
    # byte count in exactly that pattern (tools/ubench/fetch_calib.hip, profiles/r06_fetch_size_calibration.log: FETCH_SIZE x 1024 / bytes
    # requested = 1.0000 for one 64-byte segment of a 256-byte row per pass, 0.5000 for wide reads); round 5 had inferred the same factor from
    # the geometric halo amplification (profiles/r05_final_conv_tile_order.log)
    'conv3d_bf16[k3 s1 128->64 S100': (['conv3_halo_kernel<2, 1, 4, 1, 0, 2, 2, 0>'],
                                       'final conv forward (+ the SpatialSoftmax3D partials of its epilogue): 12.3 GB compulsory (2 x 4.1 GB read, 4.1 GB written)', 1.0),
    'conv3d_wgrad[k3 s1 128->64 S100]': (['wgrad_halo_kernel<2, 4, 4, 2>'],
                                         'weight gradient of the final conv (single fp16 products): 12.3 GB compulsory (x 4.1 GB + the second source 4.1 GB + dY 4.1 GB)', 2.0),
    'conv3d_bf16[k3 s1 64->128 S102': (['conv3_halo_kernel<2, 3, 4, 1, 0, 2, 2, 0>', 'conv3_halo_kernel<2, 2, 4, 1, 0, 1, 0, 1>'],
                                       'the two launches of the data gradient + padding adjoint of the final conv: d(u0) fp16x2, d(d0) fp16', 1.0),
}


def newest_pmc_profiles(root=None):
    """(fetch summary path, write summary path) of the newest rNN_vK pass under profiles/ that has both, or None"""
    import glob
    import re
    root = root or os.path.join(ROOT, 'profiles')
    best = None
    for f in glob.glob(os.path.join(root, 'r*_v*_pmc_FETCH_SIZE_summary.txt')):
        m = re.match(r'r(\d+)_v(\d+)_pmc_FETCH_SIZE_summary\.txt$', os.path.basename(f))
        wf = f.replace('FETCH_SIZE', 'WRITE_SIZE')
        if m and os.path.exists(wf):
            key = (int(m.group(1)), int(m.group(2)))
            if best is None or key > best[0]:
                best = (key, f, wf)
    return None if best is None else (best[1], best[2])


def pmc_mean_per_dispatch(path):
    """kernel name (as tools/pmc_summary.py prints it, truncated to 90 characters) -> mean counter value per dispatch"""
    out = {}
    with open(path) as fh:
        for line in fh:
            parts = line.rstrip('\n').rsplit(None, 3)
            if len(parts) == 4 and parts[1].isdigit():
                try:
                    out[parts[0].strip()] = float(parts[2])
                except ValueError:
                    pass
    return out


def profile_traffic(label):
    """(HBM bytes per call of the timer label, note) from the newest committed PMC passes, or None"""
    ent = next((v for k, v in TRAFFIC_KERNELS.items() if label.startswith(k)), None)
    prof = newest_pmc_profiles()
    if ent is None or prof is None:
        return None
    fetch, write = pmc_mean_per_dispatch(prof[0]), pmc_mean_per_dispatch(prof[1])
    total, parts = 0.0, []
    for kern in ent[0]:
        f = next((v for k, v in fetch.items() if kern in k), None)
        w = next((v for k, v in write.items() if kern in k), None)
        if f is None:
            return None
        total += (ent[2] * f + (w or 0.0)) * 1024.0
        parts.append('%s: FETCH_SIZE %.2f GB raw (x %g%s), WRITE_SIZE %.2f GB' % (
            kern, f * 1024.0 / 1e9, ent[2], '' if ent[2] == 2.0 else ': 64-byte segment loads, factor measured in profiles/r06_fetch_size_calibration.log',
            (w or 0.0) * 1024.0 / 1e9))
    return total, '%s; %s; %s' % (ent[1], '; '.join(parts), os.path.basename(prof[0]).replace('_pmc_FETCH_SIZE_summary.txt', '_pmc_*'))


# HBM bytes the voxelizer chain really moves per call at configs[1] (B=16, V=100, 4 x 128 x 128 points): from the newest committed
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/bench_voxel.py (profiles/rNN_vK_voxel_p{0,2}_pmc_*.txt: 5 warm-up + 20 timed
# calls = 25 per file; sum over the chain's kernels; FETCH_SIZE x 2 as MI355X_MICROARCH.md prescribes for gfx950).  None = no profile.
VOXEL_PROFILE_CALLS = 25


def voxel_profile_traffic(incremental, root=None):
    import glob
    import re
    root = root or os.path.join(ROOT, 'profiles')
    tag = 'p2' if incremental else 'p0'
    best = None
    for f in glob.glob(os.path.join(root, 'r*_v*_voxel_%s_pmc_FETCH_SIZE.txt' % tag)):
        m = re.match(r'r(\d+)_v(\d+)_voxel_', os.path.basename(f))
        wf = f.replace('FETCH_SIZE', 'WRITE_SIZE')
        if m and os.path.exists(wf):
            key = (int(m.group(1)), int(m.group(2)))
            if best is None or key > best[0]:
                best = (key, f, wf)
    if best is None:
        return None, None

    def total(path):
        t = 0.0
        with open(path) as fh:
            for line in fh:
                parts = line.rstrip('\n').rsplit(None, 3)
                if len(parts) == 4 and parts[1].isdigit() and ('vt_' in parts[0] or 'vox_' in parts[0]):
                    t += float(parts[3])
        return t
    return (2.0 * total(best[1]) + total(best[2])) * 1024.0 / VOXEL_PROFILE_CALLS, os.path.basename(best[1]).replace('_pmc_FETCH_SIZE.txt', '_pmc_*')


def voxel_roofline(ms, alg_bytes, V, B, incremental):
    """HBM roofline of one voxelizer call.  `achieved` / `frac` are REAL bandwidth: the HBM bytes the call moves (PMC) over its
    duration -- for the training path (two persistent grids updated in place: only the cells occupied now / two steps ago are
    touched) that is far less than the algorithmic read-points + write-grid bytes of SURVEY.md 8d, so the figure priced on those
    is reported separately as `algorithmic_equivalent_gbps` (what a full-rewrite voxelizer would need to sustain to be as fast)
    and never as a fraction of the peak."""
    traffic, traffic_src = voxel_profile_traffic(incremental) if (V, B) == (100, 16) else (None, None)
    r = {'bound': 'hbm', 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'avg_launch_ms': ms, 'traffic': traffic, 'traffic_source': traffic_src,
         'algorithmic_bytes_per_launch': alg_bytes, 'algorithmic_equivalent_gbps': alg_bytes / (ms * 1e-3) / 1e9,
         'target': 'north_star: >= 0.60 of HBM peak on the algorithmic bytes = <= %.0f us per call' % (alg_bytes / (0.6 * PEAK_HBM_GBPS * 1e9) * 1e6)}
    if incremental:
        # the call does not do the priced work (it rewrites ~3 % of the grid): no fraction of peak is claimed for it
        if traffic is not None:
            r['achieved'] = traffic / (ms * 1e-3) / 1e9
            r['frac'] = r['achieved'] / PEAK_HBM_GBPS
        else:
            r['achieved'], r['frac'] = None, None
        r['note'] = ('training path: the grid lives in two persistent buffers that are UPDATED in place (reset the cells occupied two '
                     'steps ago, write the ones occupied now: ~2 x 40 B per occupied cell + ~100 B per point); achieved = PMC bytes / '
                     'time: a latency-bound chain of six small kernels, not a bandwidth figure to compare with the 60 % target -- '
                     'see voxel_scatter_stateless for the call that does write all V^3 cells')
    else:
        r['achieved'] = alg_bytes / (ms * 1e-3) / 1e9
        r['frac'] = r['achieved'] / PEAK_HBM_GBPS
        r['note'] = ('stateless call (fresh output tensor, every cell written: what VoxelGrid.coords_to_bounding_voxel_grid and act() do); '
                     'achieved = algorithmic bytes (read N*6*4 + write V^3*10*4 per sample) / time')
    return r


def voxel_stateless(agent, batch, cfg, V, B):
    """time the stateless voxelizer call (fresh grid, all V^3 cells written) on one of the bench's batches"""
    from voxactb_amd.voxel.voxel_grid import VoxelGrid
    from voxactb_amd import synthetic
    cams = cfg.rlbench.cameras
    dev = batch['%s_rgb' % cams[0]].device
    pcd = [batch['%s_point_cloud' % c][:, 0].float().contiguous() for c in cams]
    rgb = [((batch['%s_rgb' % c][:, 0].float() / 255.0) * 2.0 - 1.0).contiguous() for c in cams]
    vg = VoxelGrid(synthetic.SCENE_BOUNDS, V, dev, B, 3, pcd[0].shape[-1] * pcd[0].shape[-2] * len(cams))
    for _ in range(3):
        vg.voxelize_cameras(pcd, rgb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        g = vg.voxelize_cameras(pcd, rgb)
    e1.record()
    torch.cuda.synchronize()
    del g
    nbytes = B * (len(cams) * pcd[0].shape[-1] * pcd[0].shape[-2] * 6 * 4 + V ** 3 * 10 * 4)
    return voxel_roofline(e0.elapsed_time(e1) / n, float(nbytes), V, B, False)


MODE_DTYPE = {         # (short: the driver's record truncates long strings; the long form is `precision_note`)
    'fp32': 'f32 (exact fp32 MFMA everywhere)',
    'bf16x3': 'f32 storage/accumulate; fwd products bf16x3 (3 bf16 MFMA), attention core 1x fp16; bwd: weight grads + attention 1x fp16, data grads 2x fp16 (scaled)',
    'bf16x3+attn_x3': 'as bf16x3, attention fwd on the bf16x3 triple (VOXACTB_ATTN_KERNEL=r3: the default of rounds 3 - 5)',
    'bf16': 'bf16 MFMA, f32 accumulate/storage',
    'bf16x3/bf16': 'fwd bf16x3, bwd products plain bf16',
    'bf16x3+attn_f16': 'as bf16x3, attention fwd on 1x fp16 products (VOXACTB_ATTN_KERNEL=auto)',
    'bf16x3+direct_final': 'as bf16x3, final conv fwd + d(u0) on the direct kernels (no Winograd depth axis)',
}
MODE_NOTE = {
    'fp32': 'exact fp32 matrix cores: the reference-parity mode of the first measurements (tests/test_encoder_gpu.py, 1e-4)',
    'bf16x3': 'f32 storage / accumulate; forward products as the bf16x3 split (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16) -- except the '
              'attention core (QK^T, PV) from 2^22 score elements on, which runs ONE fp16 product per term on the pipelined kernel (round 6: its '
              'round-5 "failure" on the F5c3 digest was one max-pool tie, not arithmetic; profiles/r06_attn_variants_*.log) -- held to the '
              '1e-4 Q-value bound of the exact-fp32 mode (tests/test_c2_reference_gpu.py, test_encoder_gpu.py); backward: weight gradients '
              '(leaves) and the attention backward (pipelined kernels, csrc/flash2_bwd.hip) as single fp16 products, propagating conv / '
              'wide-linear data gradients as two fp16 products (gradient hi + lo), all with device-side power-of-two operand scales; gradients '
              'held to 0.5 % relative L2 of the float64 reference on eight batches (tests/test_grad_noise_gpu.py)',
    'bf16': 'throughput mode, NOT held to the 1e-4 Q-value bound (tests/test_bf16_mode_gpu.py: ~4e-3 on q_trans)',
    'bf16x3/bf16': 'mixed mode (VOXACTB_BWD_PRECISION=bf16): the forward keeps the 1e-4 Q-value bound, parameter gradients are '
                   'within 0.5 % of the reference (norms within 4e-3) instead of 0.2 % -- not the default',
    'bf16x3+direct_final': 'same-run A/B of round 5\'s Winograd depth axis (DESIGN.md 4b / 5r5.7): the default precision with VOXACTB_FINAL_WINOGRAD=0 '
                           'VOXACTB_DGRAD_WINOGRAD=0 -- the `final` conv\'s forward and its propagating data gradient on round 4\'s direct LDS-halo kernels '
                           '(27 x 4 instead of 36 x 2 MFMA groups per wave and chunk)',
    'bf16x3+attn_x3': 'same-run A/B of round 6\'s default change (VOXACTB_ATTN_KERNEL=r3): the attention forward on round 3\'s bf16x3 kernel (three bf16 MFMAs '
                      'per product), everything else as the headline',
    'bf16x3+attn_f16': 'named mode, NOT the default (VOXACTB_ATTN_KERNEL=auto): the default precision with the attention FORWARD (QK^T, PV) on the '
                       'pipelined kernel with ONE fp16 product per term (csrc/flash2_fwd.hip) from 2^22 score elements on, instead of round 3\'s bf16x3 '
                       'triple.  Measured (DESIGN.md 5r5; profiles/r05_attn_f16_forward_gate.log): Q-values within 1e-4 of the reference on every fixture '
                       'at these sizes (max 8.1e-5, the default 7.0e-5), every parameter gradient inside the 0.5 % float64 gate on eight batches once the '
                       'backward is evaluated at the reference run\'s LeakyReLU choices (worst tensor 0.74 x gate, the default 0.78 x: its round-4 '
                       'rejection was the kink effect, not attention arithmetic) -- but the element gates of the F5c3 gradient digest (0.3 % of a small '
                       'tensor\'s maximum) are missed by 2 x, so it is not the default',
}
# `peak` of every matrix-core roofline = the guide's dense MFMA peak of the operand type (MI355X_MICROARCH.md): 157.3 TF/s fp32,
# 2500 TF/s bf16 / fp16.  The bf16x3 precision spends three MFMAs per product, so an ideal bf16x3 kernel tops out at 1/3 of that
# peak: reported next to `frac` as `frac_of_x3_roof` (secondary, never the headline fraction).
MODE_PEAK = {'fp32': PEAK_FP32_MFMA_TFLOPS, 'bf16x3': PEAK_BF16_MFMA_TFLOPS, 'bf16': PEAK_BF16_MFMA_TFLOPS}
MODE_PEAK_BASIS = {'fp32': 'fp32 MFMA 157.3 TF/s', 'bf16x3': 'dense bf16 MFMA 2500 TF/s (MI355X_MICROARCH.md); bf16x3 issues 3 MFMAs per product',
                   'bf16': 'dense bf16 MFMA 2500 TF/s (MI355X_MICROARCH.md)'}


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
                        'peak_basis': MODE_PEAK_BASIS[mode],
                        'launches': calls // steps, 'avg_launch_ms': ms / max(calls, 1), 'ms_per_step': ms / steps}
            if mode == 'bf16x3':
                out[key]['frac_of_x3_roof'] = tf / (peak / 3.0)
    return out


def attention_kernel_probe(dev):
    """The fused attention kernels alone at the step's self-attention size (B = 16, 8 heads, 2048 x 2048, head dim 64), random data,
    planes / preparation passes excluded for the forward and included for the backward: algorithmic TFLOP/s (4 N^2 d forward, 10 N^2 d
    backward per (batch, head)) against the dense bf16 / fp16 MFMA peak.  north_star: >= 30 % of that peak on the attention."""
    from voxactb_amd import flash
    B, H, N = 16, 8, 2048
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    q = torch.randn(B * N, H * 64, device=dev, generator=g)
    kv = torch.randn(B * N, 2 * H * 64, device=dev, generator=g)
    d_o = torch.randn(B * N, H * 64, device=dev, generator=g) * 1e-3

    def t(fn, n=6):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    out = {}
    ffl, bfl = 4.0 * B * H * N * N * 64, 10.0 * B * H * N * N * 64
    for mode in ('bf16', 'f16'):
        pl = flash.kv_planes(kv, mode)
        for p in (0.0, 0.1):
            ms = t(lambda: flash.flash2_attn_fwd(q, kv, B, H, N, N, 0.125, p, 3, mode=mode, planes=pl))
            out['fwd_%s_p%.1f%s' % (mode, p, ' (hash mask)' if p > 0 else '')] = {'ms': ms, 'achieved': ffl / ms * 1e-9, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                               'frac': ffl / ms * 1e-9 / PEAK_BF16_MFMA_TFLOPS}
    pl3 = flash._planes(kv, 2)
    o, lse = flash.flash_attn_fwd_dl(q, kv, B, H, N, N, 0.125, 0.1, 3, x3=True)
    ms = t(lambda: flash.call('vxb_flash_attn_fwd_dl', q, pl3, 2, o, lse, B, H, N, N, 64, 0.125, 0.1, 3))
    out['fwd_bf16x3_round3_kernel_p0.1 (the default forward)'] = {'ms': ms, 'achieved': ffl / ms * 1e-9, 'peak': PEAK_BF16_MFMA_TFLOPS,
                                                                  'unit': 'TFLOP/s', 'frac': ffl / ms * 1e-9 / PEAK_BF16_MFMA_TFLOPS}
    plf = flash.kv_planes(kv, 'f16')
    for gx in (False, True):
        ms = t(lambda: flash.flash2_attn_bwd(q, kv, o, d_o, lse, plf, B, H, N, N, 0.125, 0.1, 3, mode='f16', gx=gx))
        out['bwd_f16_%s_p0.1%s' % ('hi+lo' if gx else 'single', '' if gx else ' (hash mask: the default of rounds 4 - 5)')] = {
            'ms': ms, 'achieved': bfl / ms * 1e-9, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': bfl / ms * 1e-9 / PEAK_BF16_MFMA_TFLOPS}
    # the pair the step runs since round 6: the forward stores the dropout mask (drawn from a per-row LCG), the backward reads it
    o2, lse2, mask = flash.flash2_attn_fwd(q, kv, B, H, N, N, 0.125, 0.1, 3, mode='f16', planes=plf, return_mask=True)
    if mask is not None:
        ms = t(lambda: flash.flash2_attn_fwd(q, kv, B, H, N, N, 0.125, 0.1, 3, mode='f16', planes=plf, return_mask=True))
        out['fwd_f16_p0.1_stored_mask (the default forward)'] = {'ms': ms, 'achieved': ffl / ms * 1e-9, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                                               'frac': ffl / ms * 1e-9 / PEAK_BF16_MFMA_TFLOPS}
        ms = t(lambda: flash.flash2_attn_bwd(q, kv, o2, d_o, lse2, plf, B, H, N, N, 0.125, 0.1, 3, mode='f16', gx=False, drop_mask=mask))
        out['bwd_f16_single_p0.1_stored_mask (the default backward)'] = {'ms': ms, 'achieved': bfl / ms * 1e-9, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                                                        'frac': bfl / ms * 1e-9 / PEAK_BF16_MFMA_TFLOPS}
    return out


def reference_digest_check(dev, mode, V, depth, latents, HW, fixture, force_wide=False):
    """Forward of the engine in precision `mode` on the seeded batch of a reference fixture at BASELINE.json configs[1] geometry (F5: B = 1;
    f5gb8: B = 8) with name-hashed weights, against the numbers the REFERENCE produced for it.  None when the bench runs at another size.
    force_wide: dispatch the 128 x 512-tile kernels of the B = 16 timed region for this smaller batch too (ops.set_wide_min_rows)."""
    import numpy as np
    from voxactb_amd import ops, synthetic
    from voxactb_amd.agents.peract_bc.perceiver_lang_io import PerceiverVoxelLangEncoder
    from voxactb_amd.voxel.voxel_grid import VoxelGrid
    path = os.path.join(ROOT, 'tests', 'golden', fixture + '.npz')
    if not os.path.exists(path):
        return None
    g = np.load(path, allow_pickle=False)
    if (V, depth, latents, HW) != (int(g['cfg_V']), int(g['cfg_depth']), int(g['cfg_latents']), int(g['cfg_H'])):
        return None
    Bf = int(g['cfg_B'])
    old_min = ops.WIDE_MIN_M
    if force_wide:
        ops.set_wide_min_rows(1024)
    try:
        return _digest_check(dev, mode, V, depth, latents, HW, fixture, g, Bf, force_wide or Bf * latents >= old_min)
    finally:
        if force_wide:
            ops.set_wide_min_rows(old_min)


def _digest_check(dev, mode, V, depth, latents, HW, fixture, g, Bf, wide):
    import numpy as np
    from voxactb_amd import synthetic
    from voxactb_amd.agents.peract_bc.perceiver_lang_io import PerceiverVoxelLangEncoder
    from voxactb_amd.voxel.voxel_grid import VoxelGrid
    T = lambda x: torch.from_numpy(np.asarray(x))       # noqa: E731
    cams = synthetic.CAMERAS4[:int(g['cfg_ncam'])]
    enc = PerceiverVoxelLangEncoder(depth=depth, iterations=1, voxel_size=V, initial_dim=10, low_dim_size=int(g['cfg_low_dim']),
                                    num_latents=latents, voxel_patch_size=int(g['cfg_k']), voxel_patch_stride=int(g['cfg_s']),
                                    activation='lrelu', input_dropout=0.0, attn_dropout=0.0, decoder_dropout=0.0)
    enc.load_state_dict(synthetic.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, 0), strict=False)
    enc = enc.to(dev)
    rs = synthetic.make_replay_sample(Bf, cams, (HW, HW), V, int(g['cfg_low_dim']), seed=1)
    rs = {k: (v[:, 0] if v.dim() > 2 else v) for k, v in rs.items()}
    rs = {k: ((v.float() / 255.0) * 2.0 - 1.0 if 'rgb' in k else v.float()) for k, v in rs.items()}
    vg = VoxelGrid(synthetic.SCENE_BOUNDS, V, dev, Bf, 3, HW * HW * len(cams))
    grid = vg.voxelize_cameras([rs['%s_point_cloud' % c].to(dev) for c in cams], [rs['%s_rgb' % c].to(dev) for c in cams])
    occ = torch.nonzero((grid[..., -1] > 0).reshape(-1))[:, 0].int().cpu()
    eng = enc.engine()
    eng.precision = mode
    outs, _ = eng.forward(grid, rs['low_dim_state'].to(dev), rs['lang_token_embs'].to(dev), training=False, save=False)
    flat = outs[0].reshape(Bf, -1).float().cpu()
    sidx = T(g['q_trans_sample_idx']).long()
    res = {'what': 'engine forward in %s vs the reference digest tests/golden/%s.npz (V=%d, depth %d, %d latents, '
                   'B=%d, name-hashed weights): max abs error' % (mode, fixture, V, depth, latents, Bf),
           'linear_layer_dispatch': ('the timed region\'s: 128 x 512-tile GEMM + fused GEGLU epilogue (gemm_wide.hip)' if wide else
                                     '128 x 128 / 128 x 64 tiles (fewer than ops.WIDE_MIN_M rows)'),
           'voxel_occupancy_bit_exact': bool(torch.equal(occ, T(g['grid_occ_flat']))),
           'q_trans_argmax_equal': bool(torch.equal(flat.argmax(1), T(g['q_trans_argmax']))),
           'q_trans_4096_samples': float((flat[:, sidx] - T(g['q_trans_sample'])).abs().max()),
           'q_trans_top16': float((torch.gather(flat, 1, T(g['q_trans_top_idx']).long()) - T(g['q_trans_top_vals'])).abs().max()),
           'q_trans_logsumexp': float((torch.logsumexp(flat.double(), 1) - T(g['q_trans_lse'])).abs().max()),
           'rot_grip': float((outs[1].float().cpu() - T(g['rot_grip'])).abs().max()),
           'collision': float((outs[2].float().cpu() - T(g['collision'])).abs().max()),
           'q_trans_abs_max': float(flat.abs().max()), 'bound': 1e-4}
    res['within_bound'] = bool(max(res[k] for k in ('q_trans_4096_samples', 'q_trans_top16', 'q_trans_logsumexp', 'rot_grip',
                                                    'collision')) < 1e-4 and res['voxel_occupancy_bit_exact'])
    del enc, eng, outs, grid
    torch.cuda.empty_cache()
    return res


def _host_ram_gb():
    try:
        with open('/proc/meminfo') as fh:
            for line in fh:
                if line.startswith('MemTotal:'):
                    return float(line.split()[1]) / 1e6
    except OSError:
        pass
    return None


def cpu_baseline(a, cfg):
    """The oracle (CPU restatement of the reference path, kind 'port') timed on this host.  Bounded sample: ONE replay
    sample (B=1) of the same workload through voxelize + forward + losses + backward + LAMB, and the voxelizer alone at the full
    batch; `value` extrapolates a B=16 step as 16 x the sample.  --cpu-full-batch (not in the default run: ~2 minutes and ~45 GB of
    host RAM) also times ONE real B=16 step; the result of such a run is committed under profiles/."""
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
    from oracle import voxel_grid as ovox

    def batch_of(n):
        rs = synthetic.make_replay_sample(n, cfg.rlbench.cameras, (a.image, a.image), V, 4, seed=7)
        r = {k: (v[:, 0] if v.dim() > 2 else v) for k, v in rs.items()}
        r = {k: ((v.float() / 255.0) * 2.0 - 1.0 if 'rgb' in k else v.float()) for k, v in r.items()}
        return dict(pcd=[r['%s_point_cloud' % c] for c in cfg.rlbench.cameras], rgb=[r['%s_rgb' % c] for c in cfg.rlbench.cameras],
                    proprio=r['low_dim_state'], lang_token_embs=r['lang_token_embs'], bounds=torch.tensor([synthetic.SCENE_BOUNDS]),
                    trans=r['trans_action_indicies'], rot_grip=r['rot_grip_action_indicies'], ignore_collisions=r['ignore_collisions'])
    bt = batch_of(1)
    t0 = time.perf_counter()
    ovox.voxelize(*ovox.flatten_cameras(bt['pcd'], bt['rgb']), bt['bounds'], V)       # the voxelizer leg on its own
    dt_vox = time.perf_counter() - t0
    t0 = time.perf_counter()
    oagent.train_steps(ow.hashed_state_dict(shapes, 0), [bt], V, 1, depth=a.depth, voxel_patch_stride=s)
    dt = time.perf_counter() - t0
    out = {'value': 1.0 / (dt * a.batch), 'unit': 'steps/s', 'cores': ncores, 'kind': 'port', 'extrapolated': True,
           'sample': 'one replay sample (B=1) of the same workload through the CPU oracle (voxelize + fwd + 6 CE + bwd + LAMB) '
                     'took %.1f s; `value` EXTRAPOLATES a B=%d step as %dx that (a real B=%d step: --cpu-full-batch, ~45 GB of host RAM and '
                     'minutes; profiles/r06_cpu_full_batch.json)' % (dt, a.batch, a.batch, a.batch),
           'seconds_per_sample': dt, 'voxelize_seconds_per_sample': dt_vox, 'host_ram_gb': _host_ram_gb(),
           'voxelize_note': 'scatter-mean voxelization of one sample (4 x 128 x 128 points -> %d^3 grid) on the same threads' % V}
    # the voxelizer at the FULL batch (cheap: ~B x the sample)
    btB = batch_of(a.batch)
    t0 = time.perf_counter()
    ovox.voxelize(*ovox.flatten_cameras(btB['pcd'], btB['rgb']), btB['bounds'], V)
    out['voxelize_seconds_per_batch'] = time.perf_counter() - t0
    out['voxelize_batches_per_s'] = 1.0 / out['voxelize_seconds_per_batch']
    if a.cpu_full_batch:
        ram = out['host_ram_gb']
        if ram is not None and ram < 64.0:
            out['full_batch'] = 'skipped: host RAM %.0f GB < 64 GB' % ram
        else:
            t0 = time.perf_counter()
            oagent.train_steps(ow.hashed_state_dict(shapes, 0), [btB], V, 1, depth=a.depth, voxel_patch_stride=s)
            dtB = time.perf_counter() - t0
            out['full_batch'] = {'seconds_per_step': dtB, 'value': 1.0 / dtB, 'unit': 'steps/s', 'batch': a.batch,
                                 'note': 'ONE real B=%d step of the oracle (not extrapolated)' % a.batch}
    return out


if __name__ == '__main__':
    main()
