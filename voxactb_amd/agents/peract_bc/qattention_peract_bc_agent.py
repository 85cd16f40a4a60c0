"""QFunction and QAttentionPerActBCAgent -- drop-ins for the reference classes of the same names
(reference: peract/agents/peract_bc/qattention_peract_bc_agent.py:31-135 and :138-880).

Same constructor arguments, `build / update / act / update_summaries / act_summaries / load_weights / load_weight /
save_weights`, same replay-sample keys, same returned dictionaries, same checkpoint key names
(`_qnet.module.<param>` when training, `_qnet.<param>` in eval, agent :848-849).

What differs underneath (MI355X-first):
  * voxelization + camera flatten = one HIP voxelizer call on the planar camera tensors;
  * the Q-network forward AND backward are explicit HIP kernel sequences (PerceiverEngine), no autograd graph;
  * the six cross-entropy terms are fused log-softmax-NLL kernels on integer labels -- no 128 MB int64 one-hot
    clones and no per-sample Python loops (agent :519-545), argmax comes out of the same pass;
  * gradients live in one flat buffer: one RCCL all-reduce (`backend="nccl"` on ROCm) instead of DDP-over-gloo
    buckets (agent :50-54, run_seed_fn.py:34), and one fused multi-tensor LAMB (helpers/optim/lamb.py);
  * no device->host sync inside update(): the loss stays a device tensor until the runner calls .item().
"""
import logging
import os
from typing import List

import numpy as np
import torch
import torch.nn as nn

from ... import ops
from ..._lib import VoxactbHipError
from ...flat_params import FlatParams
from ...helpers.optim.adam import Adam
from ...helpers.optim.lamb import Lamb
from ...helpers.optim.schedule import CosineWithHardRestarts
from ...voxel.augmentation import se3_augmentation_plan, se3_augmentation_plan_2robots
from ...voxel.voxel_grid import VoxelGrid
from ...yarr_agent import Agent, ActResult, ScalarSummary, HistogramSummary, Summary

NAME = 'QAttentionAgent'


class _ModuleShim(nn.Module):
    """Gives the training-mode Q-network the `module.` name level DistributedDataParallel adds upstream
    (agent :50-54), so checkpoints are interchangeable.  The gradient exchange itself is FlatParams.all_reduce_grads."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


class QFunction(nn.Module):

    def __init__(self, perceiver_encoder: nn.Module, voxelizer: VoxelGrid, bounds_offset: float,
                 rotation_resolution: float, device, training, arm_pred_loss):
        super(QFunction, self).__init__()
        self._rotation_resolution = rotation_resolution
        self._voxelizer = voxelizer
        self._bounds_offset = bounds_offset
        enc = perceiver_encoder.to(device)
        self._arm_pred_loss = arm_pred_loss
        self._is_training = training
        self._qnet = _ModuleShim(enc) if training else enc
        self._device = device

    @property
    def encoder(self):
        return self._qnet.module if self._is_training else self._qnet

    def _argmax_3d(self, tensor_orig):
        """agent :57-63 (the index arithmetic assumes d == h == w, as upstream does)."""
        b, c, d, h, w = tensor_orig.shape
        flat = tensor_orig.reshape(b * c, -1).contiguous()
        _, _, idxs = ops.ce_big(flat, torch.zeros(b * c, dtype=torch.int32, device=flat.device))
        idxs = idxs.long().view(b, c)
        t = torch.div(idxs, h, rounding_mode='trunc')
        return torch.cat([torch.div(t, d, rounding_mode='trunc'), t % w, idxs % w], 1)

    def choose_highest_action(self, q_trans, q_rot_grip, q_collision):
        """agent :65-80."""
        coords = self._argmax_3d(q_trans)
        rot_and_grip_indicies = None
        ignore_collision = None
        if q_rot_grip is not None:
            n = int(360 // self._rotation_resolution)
            b = q_rot_grip.shape[0]
            lab = torch.zeros((b, 4), dtype=torch.int32, device=q_rot_grip.device)
            _, pred = ops.ce_rows(q_rot_grip.contiguous(), [(0, n), (n, n), (2 * n, n), (3 * n, q_rot_grip.shape[1] - 3 * n)], lab)
            rot_and_grip_indicies = pred.long()
            qc = q_collision[:, -2:].contiguous()
            _, predc = ops.ce_rows(qc, [(0, 2)], torch.zeros((b, 1), dtype=torch.int32, device=qc.device))
            ignore_collision = predc.long()
        return coords, rot_and_grip_indicies, ignore_collision

    def voxelize(self, rgb_pcd, pcd, bounds, xform=None):
        """agent :85-100: flatten cameras + VoxelGrid, fused (and, with `xform`, the SE(3) augmentation's rigid transform
        of the clouds, augmentation.py:36-62).  Returns the channels-last grid [B,V,V,V,10]."""
        rgb = [rp[0] for rp in rgb_pcd]
        return self._voxelizer.voxelize_cameras(pcd, rgb, bounds, xform)

    def forward(self, rgb_pcd, proprio, pcd, lang_goal_emb, lang_token_embs, bounds=None, prev_bounds=None,
                prev_layer_voxel_grid=None):
        """agent :82-135 (inference path; update() drives the engine directly to keep the backward cache)."""
        grid = self.voxelize(rgb_pcd, pcd, bounds)
        voxel_grid = grid.permute(0, 4, 1, 2, 3).detach()          # channels-first VIEW, as upstream exposes it
        eng = self.encoder.engine()
        outs, _ = eng.forward(grid, proprio, lang_token_embs, training=False, save=False, lang_goal_emb=lang_goal_emb)
        if self._arm_pred_loss and self._is_training:
            return outs[0], outs[1], outs[2], voxel_grid, outs[3]
        return outs[0], outs[1], outs[2], voxel_grid


class QAttentionPerActBCAgent(Agent):

    def __init__(self, layer: int, coordinate_bounds: list, perceiver_encoder: nn.Module, camera_names: list,
                 batch_size: int, voxel_size: int, bounds_offset: float, voxel_feature_size: int, image_crop_size: int,
                 num_rotation_classes: int, rotation_resolution: float, lr: float = 0.0001, lr_scheduler: bool = False,
                 training_iterations: int = 100000, num_warmup_steps: int = 20000, trans_loss_weight: float = 1.0,
                 rot_loss_weight: float = 1.0, grip_loss_weight: float = 1.0, collision_loss_weight: float = 1.0,
                 include_low_dim_state: bool = False, image_resolution: list = None, lambda_weight_l2: float = 0.0,
                 transform_augmentation: bool = True, transform_augmentation_xyz: list = [0.0, 0.0, 0.0],
                 transform_augmentation_rpy: list = [0.0, 0.0, 180.0], transform_augmentation_rot_resolution: int = 5,
                 optimizer_type: str = 'adam', num_devices: int = 1, crop_target_obj_voxel: bool = False, wandb_run=None,
                 arm_pred_loss: bool = False, arm_loss_weight: float = 1.0, randomizations_crop_point: bool = False):
        self._layer = layer
        if type(coordinate_bounds[0]) is float:
            self._coordinate_bounds = coordinate_bounds
        else:
            self._coordinate_bounds = coordinate_bounds[0]     # multi task: overwritten in update/act anyway (:187-190)
        self._perceiver_encoder = perceiver_encoder
        self._voxel_feature_size = voxel_feature_size
        self._bounds_offset = bounds_offset
        self._image_crop_size = image_crop_size
        self._lr = lr
        self._lr_scheduler = lr_scheduler
        self._training_iterations = training_iterations
        self._num_warmup_steps = num_warmup_steps
        self._trans_loss_weight = trans_loss_weight
        self._rot_loss_weight = rot_loss_weight
        self._grip_loss_weight = grip_loss_weight
        self._collision_loss_weight = collision_loss_weight
        self._include_low_dim_state = include_low_dim_state
        self._image_resolution = image_resolution or [128, 128]
        self._voxel_size = voxel_size
        self._camera_names = camera_names
        self._num_cameras = len(camera_names)
        self._batch_size = batch_size
        self._lambda_weight_l2 = lambda_weight_l2
        self._transform_augmentation = transform_augmentation
        self._transform_augmentation_xyz = torch.from_numpy(np.array(transform_augmentation_xyz))
        self._transform_augmentation_rpy = transform_augmentation_rpy
        self._transform_augmentation_rot_resolution = transform_augmentation_rot_resolution
        self._optimizer_type = optimizer_type
        self._num_devices = num_devices
        self._num_rotation_classes = num_rotation_classes
        self._rotation_resolution = rotation_resolution
        self._crop_target_obj_voxel = crop_target_obj_voxel
        self._wandb_run = wandb_run
        self._arm_pred_loss = arm_pred_loss
        self._arm_loss_weight = arm_loss_weight
        self._randomizations_crop_point = randomizations_crop_point
        self._name = NAME + '_layer' + str(self._layer)
        self._text_encoder = None
        if any(abs(w - 1.0) > 0 for w in (trans_loss_weight, rot_loss_weight, grip_loss_weight, collision_loss_weight,
                                          arm_loss_weight)):
            self._loss_weights = (trans_loss_weight, rot_loss_weight, grip_loss_weight, collision_loss_weight, arm_loss_weight)
        else:
            self._loss_weights = None

    # ------------------------------------------------------------------------------------------------------------ build
    def build(self, training: bool, device: torch.device = None):
        self._training = training
        self._device = device
        if device is None:
            device = torch.device('cpu')
        dev = torch.device('cuda:%d' % device) if isinstance(device, int) else torch.device(device)
        if dev.type != 'cuda':
            raise VoxactbHipError('voxactb_amd agents run on a HIP device only (got %s); there is no CPU fallback' % dev)
        self._dev = dev
        # the C-ABI launches go to the CURRENT HIP device: run_seed_fn.py / OfflineTrainRunner pass `device=rank` and never
        # call set_device themselves (reference modules follow their tensors; raw launches do not)
        torch.cuda.set_device(dev)
        self._voxelizer = VoxelGrid(coord_bounds=self._coordinate_bounds, voxel_size=self._voxel_size, device=dev,
                                    batch_size=self._batch_size if training else 1, feature_size=self._voxel_feature_size,
                                    max_num_coords=int(np.prod(self._image_resolution)) * self._num_cameras,
                                    # training: the grid lives for one step (it is returned in `prev_layer_voxel_grid` and
                                    # read by update_summaries), so two buffers updated in place in turn are enough
                                    persistent=2 if training else 0)
        self._q = self._make_q(dev, training).to(dev).train(training)
        self._coordinate_bounds = torch.tensor(self._coordinate_bounds, device=dev).unsqueeze(0)
        if self._training:
            self._arena = FlatParams(self._q, dev)
            self._arena.broadcast_weights(0)            # DDP's wrap-time parameter broadcast (agent :50-54)
            # gradient-exchange buckets: slices of the flat gradient buffer in the order backward completes them
            shim = '_qnet.module.'
            for bname, prefixes in self._q.encoder.engine().grad_buckets():
                present = [shim + p for p in prefixes if any(n.startswith(shim + p) for n in self._arena.names)]
                if present:
                    self._arena.bucket(bname, present)
            if not self._arena.buckets_cover_everything():
                raise VoxactbHipError('gradient buckets do not tile the parameter arena')
            if self._optimizer_type == 'lamb':
                self._optimizer = Lamb(self._q.parameters(), lr=self._lr, weight_decay=self._lambda_weight_l2,
                                       betas=(0.9, 0.999), adam=False)
                self._optimizer.attach(self._arena)
            elif self._optimizer_type == 'adam':                        # agent :263-268
                self._optimizer = Adam(self._q.parameters(), lr=self._lr, weight_decay=self._lambda_weight_l2)
                self._optimizer.attach(self._arena)
            else:
                raise Exception('Unknown optimizer type')
            if self._lr_scheduler:                                      # agent :273-279
                self._scheduler = CosineWithHardRestarts(self._optimizer, num_warmup_steps=self._num_warmup_steps,
                                                         num_training_steps=self._training_iterations,
                                                         num_cycles=self._training_iterations // 10000)
            logging.info('# Q Params: %d' % sum(p.numel() for name, p in self._q.named_parameters()
                                                 if p.requires_grad and 'clip' not in name))
        else:
            for param in self._q.parameters():
                param.requires_grad = False
            # evaluation: the weights only change through load_weights() -- act() keeps their prepared forms (perceiver_lang_io.py: freeze_weight_prep)
            self._q.encoder.engine().freeze_weight_prep = True

    def _make_q(self, dev, training):
        return QFunction(self._perceiver_encoder, self._voxelizer, self._bounds_offset, self._rotation_resolution, dev,
                         training, self._arm_pred_loss)

    def set_text_encoder(self, fn):
        """fn(tokens [77] long) -> (lang_goal_emb [1,1024], lang_token_embs [1,77,512]).  Upstream loads CLIP RN50 here
        (agent :324-328); its weights (`data/clip_rn50.pth`) are not part of either repository."""
        self._text_encoder = fn

    # ------------------------------------------------------------------------------------------------------------ helpers
    def _preprocess_inputs(self, replay_sample):
        obs, pcds = [], []
        for n in self._camera_names:
            rgb = replay_sample['%s_rgb' % n]
            pcd = replay_sample['%s_point_cloud' % n]
            obs.append([rgb, pcd])
            pcds.append(pcd)
        return obs, pcds

    _act_preprocess_inputs = _preprocess_inputs

    def _softmax_q_trans(self, q):
        flat = q.reshape(q.shape[0], -1).clone()
        ld = flat.shape[1]
        if ld % 4:
            raise VoxactbHipError('V^3 must be a multiple of 4')
        ops.softmax_rows(flat, flat.shape[0], ld, ld)
        return flat.reshape(q.shape)

    def _softmax_q_rot_grip(self, q_rot_grip):
        n = self._num_rotation_classes
        out = []
        for a, b in ((0, n), (n, 2 * n), (2 * n, 3 * n), (3 * n, q_rot_grip.shape[1])):
            w = b - a
            buf = torch.zeros((q_rot_grip.shape[0], (w + 3) & ~3), dtype=torch.float32, device=q_rot_grip.device)
            buf[:, :w] = q_rot_grip[:, a:b]
            ops.softmax_rows(buf, buf.shape[0], w, buf.shape[1])
            out.append(buf[:, :w])
        return torch.cat(out, dim=1)

    def _softmax_ignore_collision(self, q_collision):
        buf = torch.zeros((q_collision.shape[0], 4), dtype=torch.float32, device=q_collision.device)
        buf[:, :2] = q_collision
        ops.softmax_rows(buf, buf.shape[0], 2, 4)
        return buf[:, :2]

    def _check_se3_status(self):
        """augmentation.py:119-120 raises after 100 failed attempts, BEFORE the forward pass.  The device kernel flags the same
        condition (and poisons that step's labels, so its loss is NaN); the optimizer kernels read the flag on the device and
        make that step a no-op (`_gate_step`), so weights and moments are exactly what they were -- the exception itself is
        raised here, at the start of the next update() / in save_weights(), without a mid-step sync."""
        st = getattr(self, '_se3_status', None)
        if st is not None:
            self._se3_status = None
            self._optimizer.skip_flag = None
            if int(st.item()) < 0:
                # that step was a no-op on the device: take the host-side step counters back too (bias correction and the reference's
                # `state[p]['step']` must match the moments), and drop the delayed fp16 operand scales its all-NaN gradients reported
                opt = self._optimizer
                opt.steps = max(0, opt.steps - 1)
                for p_ in opt.state:
                    if 'step' in opt.state[p_]:
                        opt.state[p_]['step'] = opt.steps
                self._q.encoder.engine()._grad_scales.clear()
                raise Exception('Failing to perturb action and keep it within bounds.')

    def _gate_step(self):
        """hand the SE(3) status word of this step to the optimizer kernels (negative -> no-op); with several ranks every rank
        must take the same decision, so the words are MIN-reduced first (4 bytes, started here, overlapped with the step)."""
        st = getattr(self, '_se3_status', None)
        self._status_work = None
        if st is None:
            return
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            self._status_work = torch.distributed.all_reduce(st, op=torch.distributed.ReduceOp.MIN, async_op=True)
        self._optimizer.skip_flag = st

    def _step_optimizer(self):
        if getattr(self, '_status_work', None) is not None:
            self._status_work.wait()
            self._status_work = None
        self._optimizer.step()

    # ------------------------------------------------------------------------------------------------------------ update
    def update(self, step: int, replay_sample: dict) -> dict:
        action_trans = replay_sample['trans_action_indicies'][:, self._layer * 3:self._layer * 3 + 3].int()
        action_rot_grip = replay_sample['rot_grip_action_indicies'].int()
        action_gripper_pose = replay_sample['gripper_pose']
        action_ignore_collisions = replay_sample['ignore_collisions'].int()
        action_label = replay_sample.get('label', None)
        lang_goal_emb = replay_sample['lang_goal_emb'].float()
        lang_token_embs = replay_sample['lang_token_embs'].float()
        prev_layer_voxel_grid = replay_sample.get('prev_layer_voxel_grid', None)
        prev_layer_bounds = replay_sample.get('prev_layer_bounds', None)
        device = self._dev

        if self._crop_target_obj_voxel:                                        # agent :431-449
            self._coordinate_bounds = replay_sample['target_object_scene_bounds']
            if self._randomizations_crop_point:
                r = [np.random.uniform(low=-0.05, high=0.05) for _ in range(3)]
                for a in range(3):
                    self._coordinate_bounds[:, a] += r[a]
                    self._coordinate_bounds[:, a + 3] += r[a]
        bounds = self._coordinate_bounds.to(device)
        if self._layer > 0:
            cp = replay_sample['attention_coordinate_layer_%d' % (self._layer - 1)]
            bounds = torch.cat([cp - self._bounds_offset, cp + self._bounds_offset], dim=1)
        self._q.encoder.engine().prepare_step()          # (weight forms of this step first: the GPU has work while the host enqueues the voxelizer chain)
        proprio = replay_sample['low_dim_state'] if self._include_low_dim_state else None
        obs, pcd = self._preprocess_inputs(replay_sample)
        bs = pcd[0].shape[0]

        xform = None
        if self._transform_augmentation:                                       # agent :469-483
            self._check_se3_status()               # last step's retry budget (no sync: that step has long finished)
            action_trans, action_rot_grip, xform, self._se3_status = se3_augmentation_plan(
                action_gripper_pose.to(device), action_rot_grip.to(device), bounds, self._layer,
                self._transform_augmentation_xyz, self._transform_augmentation_rpy,
                self._transform_augmentation_rot_resolution, self._voxel_size, self._rotation_resolution, device)
            self._gate_step()

        # forward (agent :486-508): voxelize (the augmentation's rigid transform rides on the point load) + encoder,
        # keeping the backward cache
        grid = self._q.voxelize(obs, pcd, bounds, xform)
        voxel_grid = grid.permute(0, 4, 1, 2, 3).detach()
        eng = self._q.encoder.engine()
        outs, cache = eng.forward(grid, proprio, lang_token_embs, training=True, save=True, lang_goal_emb=lang_goal_emb)
        q_trans, q_rot_grip, q_collision = outs[0], outs[1], outs[2]
        arm_out = outs[3] if self._arm_pred_loss else None

        # losses + argmax (agent :511-578) -- integer labels, one fused kernel per head group
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size()
        gscale = 1.0 / (bs * world)
        V = self._voxel_size
        n = self._num_rotation_classes
        at = action_trans.to(device).long()
        flat_label = ((at[:, 0] * V + at[:, 1]) * V + at[:, 2]).int()
        w = self._loss_weights or (1.0, 1.0, 1.0, 1.0, 1.0)
        dq = torch.empty((bs, V ** 3), dtype=torch.float32, device=device)
        l_trans, _, amax = ops.ce_big(q_trans.view(bs, -1), flat_label, dq, gscale * w[0])
        o = cache['o']
        labs = torch.cat([action_rot_grip.to(device).int(), action_ignore_collisions.to(device).int()[:, :1]], dim=1).contiguous()
        d_o = torch.empty_like(o)
        l_heads, pred = ops.ce_rows(o, [(0, n), (n, n), (2 * n, n), (3 * n, 2), (3 * n + 2, 2)], labs, d_o, gscale)
        if self._loss_weights is not None:
            d_o[:, :3 * n] *= w[1]
            d_o[:, 3 * n:3 * n + 2] *= w[2]
            d_o[:, 3 * n + 2:] *= w[3]
        q_rot_loss = l_heads[:, 0] + l_heads[:, 1] + l_heads[:, 2]
        q_grip_loss, q_collision_loss = l_heads[:, 3], l_heads[:, 4]
        combined = l_trans * w[0] + q_rot_loss * w[1] + q_grip_loss * w[2] + q_collision_loss * w[3]
        d_arm, q_arm_loss = None, None
        if self._arm_pred_loss:
            d_arm = torch.empty_like(arm_out)
            la, _ = ops.ce_rows(arm_out, [(0, 2)], action_label.to(device).int()[:, :1].contiguous(), d_arm, gscale * w[4])
            q_arm_loss = la[:, 0]
            combined = combined + q_arm_loss * w[4]
        total_loss = combined.mean()

        # backward + exchange + optimizer (agent :580-582)
        self._arena.zero_grad()
        # every bucket's all-reduce (RCCL over xGMI) starts the moment backward has enqueued its last gradient kernel and runs
        # next to the rest of the backward pass; the optimizer waits for all of them (DDP's bucketed overlap, agent :50-54)
        eng.backward(cache, dq, d_o, d_arm, on_bucket_ready=self._arena.reduce_bucket)
        self._arena.finish_reduce()
        self._step_optimizer()

        coords = torch.stack([torch.div(torch.div(amax, V, rounding_mode='trunc'), V, rounding_mode='trunc'),
                              torch.div(amax, V, rounding_mode='trunc') % V, amax % V], 1).long()
        self._summaries = {
            'losses/total_loss': total_loss,
            'losses/trans_loss': l_trans.mean(),
            'losses/rot_loss': q_rot_loss.mean(),
            'losses/grip_loss': q_grip_loss.mean(),
            'losses/collision_loss': q_collision_loss.mean(),
        }
        if self._arm_pred_loss:
            self._summaries['losses/arm_loss'] = q_arm_loss.mean()
        if self._lr_scheduler:                                          # agent :595-597
            self._scheduler.step()
            self._summaries['learning_rate'] = self._scheduler.get_last_lr()[0]
        self._vis_voxel_grid = voxel_grid[0]
        self._vis_translation_qvalue = None          # computed lazily in update_summaries (softmax over V^3)
        self._vis_q_trans0 = q_trans[0:1]
        self._vis_max_coordinate = coords[0]
        self._vis_gt_coordinate = action_trans[0]
        self._last_pred = (coords, pred)

        prev_layer_voxel_grid = [voxel_grid] if prev_layer_voxel_grid is None else prev_layer_voxel_grid + [voxel_grid]
        if prev_layer_bounds is None:
            prev_layer_bounds = [self._coordinate_bounds.repeat(bs, 1)]
        else:
            prev_layer_bounds = prev_layer_bounds + [bounds]
        return {'total_loss': total_loss, 'prev_layer_voxel_grid': prev_layer_voxel_grid,
                'prev_layer_bounds': prev_layer_bounds}

    # ------------------------------------------------------------------------------------------------------------ act
    def act(self, step: int, observation: dict, deterministic=False, which_arm=None, new_scene_bounds=None,
            dominant_assitive_policy=False, ep_number=0, is_real_robot=False) -> ActResult:
        deterministic = True
        if new_scene_bounds is not None:
            self._coordinate_bounds = torch.tensor(new_scene_bounds, device=self._dev).unsqueeze(0)
        bounds = self._coordinate_bounds
        prev_layer_voxel_grid = observation.get('prev_layer_voxel_grid', None)
        prev_layer_bounds = observation.get('prev_layer_bounds', None)
        if 'lang_token_embs' in observation:
            lang_goal_emb = observation['lang_goal_emb']
            lang_token_embs = observation['lang_token_embs']
            while lang_token_embs.dim() > 3:
                lang_token_embs = lang_token_embs[0]
        else:
            key = {'multiarm_left': 'lang_goal_tokens_left', 'multiarm_right': 'lang_goal_tokens_right'}.get(which_arm, 'lang_goal_tokens')
            if self._text_encoder is None:
                raise VoxactbHipError('act(): no text encoder set (set_text_encoder) and no precomputed lang_token_embs '
                                      'in the observation; upstream loads CLIP RN50 weights that are not in the tree')
            tokens = observation.get(key, None).long()
            with torch.no_grad():
                lang_goal_emb, lang_token_embs = self._text_encoder(tokens[0].to(self._dev))
        res = (bounds[:, 3:] - bounds[:, :3]) / self._voxel_size
        proprio = None
        if self._include_low_dim_state:                                         # agent :672-681
            if dominant_assitive_policy:
                proprio = torch.cat((observation['low_dim_state_left_arm'][:, :, :3], observation['low_dim_state_right_arm']), 2)
            elif which_arm in ('right', 'multiarm_right'):
                proprio = observation['low_dim_state_right_arm']
            elif which_arm in ('left', 'multiarm_left'):
                proprio = observation['low_dim_state_left_arm']
            else:
                proprio = observation['low_dim_state']
        obs, pcd = self._act_preprocess_inputs(observation)
        obs = [[o[0][0].to(self._dev), o[1][0].to(self._dev)] for o in obs]
        proprio = proprio[0].to(self._dev)
        pcd = [p[0].to(self._dev) for p in pcd]
        lang_token_embs = lang_token_embs.to(self._dev).float()
        bounds = torch.as_tensor(bounds, device=self._dev)

        q_trans, q_rot_grip, q_ignore_collisions, vox_grid = self._q(obs, proprio, pcd, lang_goal_emb, lang_token_embs,
                                                                     bounds, prev_layer_bounds, prev_layer_voxel_grid)
        q_trans = self._softmax_q_trans(q_trans)
        q_rot_grip = self._softmax_q_rot_grip(q_rot_grip)
        q_ignore_collisions = self._softmax_ignore_collision(q_ignore_collisions)
        coords, rot_and_grip_indicies, ignore_collisions = self._q.choose_highest_action(q_trans, q_rot_grip, q_ignore_collisions)
        rot_grip_action = rot_and_grip_indicies
        ignore_collisions_action = ignore_collisions.int()
        coords = coords.int()
        attention_coordinate = bounds[:, :3] + res * coords + res / 2
        prev_layer_voxel_grid = [vox_grid] if prev_layer_voxel_grid is None else prev_layer_voxel_grid + [vox_grid]
        prev_layer_bounds = [bounds] if prev_layer_bounds is None else prev_layer_bounds + [bounds]
        observation_elements = {'attention_coordinate': attention_coordinate, 'prev_layer_voxel_grid': prev_layer_voxel_grid,
                                'prev_layer_bounds': prev_layer_bounds}
        info = {'voxel_grid_depth%d' % self._layer: vox_grid, 'q_depth%d' % self._layer: q_trans,
                'voxel_idx_depth%d' % self._layer: coords}
        self._act_voxel_grid = vox_grid[0]
        self._act_max_coordinate = coords[0]
        self._act_qvalues = q_trans[0].detach()
        return ActResult((coords, rot_grip_action, ignore_collisions_action), observation_elements=observation_elements,
                         info=info)

    # ------------------------------------------------------------------------------------------------------------ summaries / io
    def update_summaries(self) -> List[Summary]:
        summaries = []       # the voxel rendering (pyrender/OpenGL, agent :791-803) is out of scope; upstream also skips it headless
        wandb_dict = {}
        for n, v in self._summaries.items():
            summaries.append(ScalarSummary('%s/%s' % (self._name, n), v))
            if self._wandb_run is not None:
                wandb_dict['%s/%s' % (self._name, n)] = v
        for tag, param in self._q.named_parameters():
            summaries.append(HistogramSummary('%s/gradient/%s' % (self._name, tag), param.grad))
            summaries.append(HistogramSummary('%s/weight/%s' % (self._name, tag), param.data))
        return summaries, wandb_dict

    def act_summaries(self) -> List[Summary]:
        return []

    def _load(self, weight_file):
        state_dict = torch.load(weight_file, map_location=self._dev)
        merged = self._q.state_dict()
        for k, v in state_dict.items():
            if not self._training:
                k = k.replace('_qnet.module', '_qnet')
            if k in merged:
                merged[k] = v
            elif '_voxelizer' not in k:
                logging.warning("key %s not found in checkpoint" % k)
        self._q.load_state_dict(merged)
        self._q.encoder.engine()._grad_scales.clear()   # delayed fp16 operand scales belong to the weights that were just replaced
        self._q.encoder.engine()._prep_sig = None       # ... and so do the prepared forms an evaluation agent keeps between act() calls
        if self._training:
            self._arena.broadcast_weights(0)            # every rank resumes from rank 0's file contents

    def load_weights(self, savedir: str):
        weight_file = os.path.join(savedir, '%s.pt' % self._name)
        self._load(weight_file)
        print("loaded weights from %s" % weight_file)

    def load_weight(self, ckpt_file: str):
        self._load(ckpt_file)
        print("loaded weights from %s" % ckpt_file)

    def save_weights(self, savedir: str):
        if getattr(self, '_optimizer', None) is not None:
            self._check_se3_status()           # the last training step's augmentation status has not been looked at yet
        torch.save(self._q.state_dict(), os.path.join(savedir, '%s.pt' % self._name))


# ======================================================================================================================
# one_policy_more_heads baseline: ONE policy predicts both arms' actions from both arms' proprioception
# ======================================================================================================================
class QFunction2Robots(QFunction):
    """reference agent :882-964."""

    def __init__(self, perceiver_encoder: nn.Module, voxelizer: VoxelGrid, bounds_offset: float, rotation_resolution: float,
                 device, training):
        super().__init__(perceiver_encoder, voxelizer, bounds_offset, rotation_resolution, device, training, False)

    def forward(self, rgb_pcd, proprio_right, proprio_left, pcd, lang_goal_emb, lang_token_embs, bounds=None, prev_bounds=None,
                prev_layer_voxel_grid=None):
        """-> (q_trans_right, q_rot_grip_right, q_collision_right, voxel_grid, q_trans_left, q_rot_grip_left,
        q_collision_left)   (agent :924-964)."""
        grid = self.voxelize(rgb_pcd, pcd, bounds)
        voxel_grid = grid.permute(0, 4, 1, 2, 3).detach()
        outs, _ = self.encoder.engine().forward(grid, proprio_right, lang_token_embs, training=False, save=False,
                                                proprio_left=proprio_left)
        return outs[0], outs[1], outs[2], voxel_grid, outs[3], outs[4], outs[5]


class QAttentionPerActBCAgent2Robots(QAttentionPerActBCAgent):
    """reference agent :966-1672.  Same constructor as upstream (no crop / arm-prediction options)."""

    def __init__(self, layer: int, coordinate_bounds: list, perceiver_encoder: nn.Module, camera_names: list,
                 batch_size: int, voxel_size: int, bounds_offset: float, voxel_feature_size: int, image_crop_size: int,
                 num_rotation_classes: int, rotation_resolution: float, lr: float = 0.0001, lr_scheduler: bool = False,
                 training_iterations: int = 100000, num_warmup_steps: int = 20000, trans_loss_weight: float = 1.0,
                 rot_loss_weight: float = 1.0, grip_loss_weight: float = 1.0, collision_loss_weight: float = 1.0,
                 include_low_dim_state: bool = False, image_resolution: list = None, lambda_weight_l2: float = 0.0,
                 transform_augmentation: bool = True, transform_augmentation_xyz: list = [0.0, 0.0, 0.0],
                 transform_augmentation_rpy: list = [0.0, 0.0, 180.0], transform_augmentation_rot_resolution: int = 5,
                 optimizer_type: str = 'adam', num_devices: int = 1, wandb_run=None):
        super().__init__(layer, coordinate_bounds, perceiver_encoder, camera_names, batch_size, voxel_size, bounds_offset,
                         voxel_feature_size, image_crop_size, num_rotation_classes, rotation_resolution, lr, lr_scheduler,
                         training_iterations, num_warmup_steps, trans_loss_weight, rot_loss_weight, grip_loss_weight,
                         collision_loss_weight, include_low_dim_state, image_resolution, lambda_weight_l2,
                         transform_augmentation, transform_augmentation_xyz, transform_augmentation_rpy,
                         transform_augmentation_rot_resolution, optimizer_type, num_devices, wandb_run=wandb_run)

    def _make_q(self, dev, training):
        return QFunction2Robots(self._perceiver_encoder, self._voxelizer, self._bounds_offset, self._rotation_resolution, dev,
                                training)

    def _arm_losses(self, q_trans, o, action_trans, action_rot_grip, action_ignore_collisions, gscale):
        """the five cross entropies of one arm (agent :1283-1363) and their gradients."""
        device, V, n = self._dev, self._voxel_size, self._num_rotation_classes
        bs = q_trans.shape[0]
        w = self._loss_weights or (1.0, 1.0, 1.0, 1.0, 1.0)
        at = action_trans.to(device).long()
        flat_label = ((at[:, 0] * V + at[:, 1]) * V + at[:, 2]).int()
        dq = torch.empty((bs, V ** 3), dtype=torch.float32, device=device)
        l_trans, _, amax = ops.ce_big(q_trans.view(bs, -1), flat_label, dq, gscale * w[0])
        labs = torch.cat([action_rot_grip.to(device).int(), action_ignore_collisions.to(device).int()[:, :1]], dim=1).contiguous()
        d_o = torch.empty_like(o)
        l_heads, pred = ops.ce_rows(o, [(0, n), (n, n), (2 * n, n), (3 * n, 2), (3 * n + 2, 2)], labs, d_o, gscale)
        if self._loss_weights is not None:
            d_o[:, :3 * n] *= w[1]
            d_o[:, 3 * n:3 * n + 2] *= w[2]
            d_o[:, 3 * n + 2:] *= w[3]
        rot = l_heads[:, 0] + l_heads[:, 1] + l_heads[:, 2]
        return dict(trans=l_trans, rot=rot, grip=l_heads[:, 3], coll=l_heads[:, 4], dq=dq, d_o=d_o, amax=amax, pred=pred,
                    combined=l_trans * w[0] + rot * w[1] + l_heads[:, 3] * w[2] + l_heads[:, 4] * w[3])

    def update(self, step: int, replay_sample: dict) -> dict:
        L = self._layer
        action_trans_right = replay_sample['trans_action_indicies_right'][:, L * 3:L * 3 + 3].int()
        action_rot_grip_right = replay_sample['rot_grip_action_indicies_right'].int()
        action_gripper_pose_right = replay_sample['gripper_pose_right']
        action_trans_left = replay_sample['trans_action_indicies_left'][:, L * 3:L * 3 + 3].int()
        action_rot_grip_left = replay_sample['rot_grip_action_indicies_left'].int()
        action_gripper_pose_left = replay_sample['gripper_pose_left']
        action_ignore_collisions = replay_sample['ignore_collisions'].int()
        lang_token_embs = replay_sample['lang_token_embs'].float()
        prev_layer_voxel_grid = replay_sample.get('prev_layer_voxel_grid', None)
        prev_layer_bounds = replay_sample.get('prev_layer_bounds', None)
        device = self._dev
        bounds = self._coordinate_bounds.to(device)
        if L > 0:
            cp = replay_sample['attention_coordinate_layer_%d' % (L - 1)]
            bounds = torch.cat([cp - self._bounds_offset, cp + self._bounds_offset], dim=1)
        proprio_right = proprio_left = None
        if self._include_low_dim_state:
            proprio_right = replay_sample['low_dim_state_right_arm']
            proprio_left = replay_sample['low_dim_state_left_arm']
        self._q.encoder.engine().prepare_step()          # (as the one-arm agent: weight forms first)
        obs, pcd = self._preprocess_inputs(replay_sample)
        bs = pcd[0].shape[0]

        xform = None
        if self._transform_augmentation:                                       # agent :1260-1281
            self._check_se3_status()
            action_trans_right, action_rot_grip_right, action_trans_left, action_rot_grip_left, xform, self._se3_status = \
                se3_augmentation_plan_2robots(
                    action_gripper_pose_right.to(device), action_rot_grip_right.to(device), action_gripper_pose_left.to(device),
                    action_rot_grip_left.to(device), bounds, L, self._transform_augmentation_xyz,
                    self._transform_augmentation_rpy, self._transform_augmentation_rot_resolution, self._voxel_size,
                    self._rotation_resolution, device)
            self._gate_step()

        grid = self._q.voxelize(obs, pcd, bounds, xform)
        voxel_grid = grid.permute(0, 4, 1, 2, 3).detach()
        eng = self._q.encoder.engine()
        outs, cache = eng.forward(grid, proprio_right, lang_token_embs, training=True, save=True, proprio_left=proprio_left)

        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size()
        gscale = 1.0 / (bs * world)
        r = self._arm_losses(outs[0], cache['o'], action_trans_right, action_rot_grip_right, action_ignore_collisions, gscale)
        l = self._arm_losses(outs[3], cache['left']['o'], action_trans_left, action_rot_grip_left, action_ignore_collisions, gscale)
        total_loss = (r['combined'] + l['combined']).mean()                   # agent :1365-1369

        self._arena.zero_grad()
        eng.backward(cache, r['dq'], r['d_o'], None, on_bucket_ready=self._arena.reduce_bucket, dq_trans_left=l['dq'],
                     d_o_left=l['d_o'])
        self._arena.finish_reduce()
        self._step_optimizer()

        V = self._voxel_size

        def coords_of(amax):
            return torch.stack([torch.div(torch.div(amax, V, rounding_mode='trunc'), V, rounding_mode='trunc'),
                                torch.div(amax, V, rounding_mode='trunc') % V, amax % V], 1).long()
        self._summaries = {
            'losses/total_loss': total_loss,
            'losses/trans_loss': (r['trans'] + l['trans']).mean(),
            'losses/rot_loss': (r['rot'] + l['rot']).mean(),
            'losses/grip_loss': (r['grip'] + l['grip']).mean(),
            'losses/collision_loss': (r['coll'] + l['coll']).mean(),
        }
        if self._lr_scheduler:
            self._scheduler.step()
            self._summaries['learning_rate'] = self._scheduler.get_last_lr()[0]
        self._vis_voxel_grid = voxel_grid[0]
        self._vis_max_coordinate_right, self._vis_gt_coordinate_right = coords_of(r['amax'])[0], action_trans_right[0]
        self._vis_max_coordinate_left, self._vis_gt_coordinate_left = coords_of(l['amax'])[0], action_trans_left[0]
        self._last_pred = (coords_of(r['amax']), r['pred'], coords_of(l['amax']), l['pred'])
        prev_layer_voxel_grid = [voxel_grid] if prev_layer_voxel_grid is None else prev_layer_voxel_grid + [voxel_grid]
        if prev_layer_bounds is None:
            prev_layer_bounds = [self._coordinate_bounds.repeat(bs, 1)]
        else:
            prev_layer_bounds = prev_layer_bounds + [bounds]
        return {'total_loss': total_loss, 'prev_layer_voxel_grid': prev_layer_voxel_grid,
                'prev_layer_bounds': prev_layer_bounds}

    def act(self, step: int, observation: dict, deterministic=False) -> ActResult:
        """agent :1457-1582: both arms' actions from one forward."""
        bounds = self._coordinate_bounds
        prev_layer_voxel_grid = observation.get('prev_layer_voxel_grid', None)
        prev_layer_bounds = observation.get('prev_layer_bounds', None)
        if 'lang_token_embs' in observation:
            lang_goal_emb = observation['lang_goal_emb']
            lang_token_embs = observation['lang_token_embs']
            while lang_token_embs.dim() > 3:
                lang_token_embs = lang_token_embs[0]
        else:
            if self._text_encoder is None:
                raise VoxactbHipError('act(): no text encoder set (set_text_encoder) and no precomputed lang_token_embs '
                                      'in the observation; upstream loads CLIP RN50 weights that are not in the tree')
            tokens = observation.get('lang_goal_tokens', None).long()
            with torch.no_grad():
                lang_goal_emb, lang_token_embs = self._text_encoder(tokens[0].to(self._dev))
        res = (bounds[:, 3:] - bounds[:, :3]) / self._voxel_size
        obs, pcd = self._act_preprocess_inputs(observation)
        obs = [[o[0][0].to(self._dev), o[1][0].to(self._dev)] for o in obs]
        proprio_right = observation['low_dim_state_right_arm'][0].to(self._dev)
        proprio_left = observation['low_dim_state_left_arm'][0].to(self._dev)
        pcd = [p[0].to(self._dev) for p in pcd]
        lang_token_embs = lang_token_embs.to(self._dev).float()
        bounds = torch.as_tensor(bounds, device=self._dev)
        qtr, qrr, qcr, vox_grid, qtl, qrl, qcl = self._q(obs, proprio_right, proprio_left, pcd, lang_goal_emb, lang_token_embs,
                                                        bounds, prev_layer_bounds, prev_layer_voxel_grid)
        out = {}
        for side, qt, qr, qc in (('right', qtr, qrr, qcr), ('left', qtl, qrl, qcl)):
            qt = self._softmax_q_trans(qt)
            coords, rot_grip, coll = self._q.choose_highest_action(qt, self._softmax_q_rot_grip(qr), self._softmax_ignore_collision(qc))
            coords = coords.int()
            out[side] = (qt, coords, rot_grip, coll.int(), bounds[:, :3] + res * coords + res / 2)
        prev_layer_voxel_grid = [vox_grid] if prev_layer_voxel_grid is None else prev_layer_voxel_grid + [vox_grid]
        prev_layer_bounds = [bounds] if prev_layer_bounds is None else prev_layer_bounds + [bounds]
        observation_elements = {'attention_coordinate_right': out['right'][4], 'attention_coordinate_left': out['left'][4],
                                'prev_layer_voxel_grid': prev_layer_voxel_grid, 'prev_layer_bounds': prev_layer_bounds}
        info = {'voxel_grid_depth%d' % self._layer: vox_grid,
                'q_depth_right%d' % self._layer: out['right'][0], 'voxel_idx_depth_right%d' % self._layer: out['right'][1],
                'q_depth_left%d' % self._layer: out['left'][0], 'voxel_idx_depth_left%d' % self._layer: out['left'][1]}
        self._act_voxel_grid = vox_grid[0]
        self._act_max_coordinate_right, self._act_qvalues_right = out['right'][1][0], out['right'][0][0].detach()
        self._act_max_coordinate_left, self._act_qvalues_left = out['left'][1][0], out['left'][0][0].detach()
        return ActResult((out['right'][1], out['right'][2], out['right'][3], out['left'][1], out['left'][2], out['left'][3]),
                         observation_elements=observation_elements, info=info)
