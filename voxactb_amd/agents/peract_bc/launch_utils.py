"""LaunchUtils API -- `create_agent(cfg)` and the replay schema of `create_replay(...)`
(reference: peract/agents/peract_bc/launch_utils.py:37-164 and :663-829).

`create_agent` accepts any object with the attribute paths hydra's DictConfig exposes upstream (cfg.method.*,
cfg.rlbench.*, cfg.replay.batch_size, cfg.ddp.num_devices, cfg.framework.*), see `default_cfg()`.
`create_replay` builds YARR's TaskUniformReplayBuffer when `yarr` is importable (the replay store is a "next" row of
SURVEY.md section 8f, not rebuilt here); `replay_schema` returns the element list without YARR.
Demo loading (`fill_replay`, :491-660) needs RLBench data and is out of scope.
"""
from types import SimpleNamespace

import numpy as np

from ...helpers.preprocess_agent import PreprocessAgent
from .perceiver_lang_io import PerceiverVoxelLangEncoder
from .qattention_peract_bc_agent import QAttentionPerActBCAgent
from .qattention_stack_agent import QAttentionStackAgent

REWARD_SCALE = 100.0
LOW_DIM_DOMINANT_ASSISTIVE_SIZE = 7
LOW_DIM_SIZE = 4


def _ns(d):
    return SimpleNamespace(**{k: (_ns(v) if isinstance(v, dict) else v) for k, v in d.items()})


def default_cfg(**over):
    """PERACT_BC.yaml + config.yaml defaults as a namespace; keyword overrides use dotted keys joined by '__'
    (e.g. method__voxel_sizes=[50], replay__batch_size=2)."""
    cfg = dict(
        method=dict(name='PERACT_BC', image_crop_size=64, bounds_offset=[0.15], voxel_sizes=[100], num_latents=2048,
                    latent_dim=512, transformer_depth=6, transformer_iterations=1, cross_heads=1, cross_dim_head=64,
                    latent_heads=8, latent_dim_head=64, pos_encoding_with_lang=True, lang_fusion_type='seq',
                    voxel_patch_size=5, voxel_patch_stride=5, final_dim=64, input_dropout=0.1, attn_dropout=0.1,
                    decoder_dropout=0.0, lr=0.0005, lr_scheduler=False, num_warmup_steps=3000, optimizer='lamb',
                    lambda_weight_l2=0.000001, trans_loss_weight=1.0, rot_loss_weight=1.0, grip_loss_weight=1.0,
                    collision_loss_weight=1.0, rotation_resolution=5, activation='lrelu',
                    transform_augmentation=dict(apply_se3=True, aug_xyz=[0.125, 0.125, 0.125], aug_rpy=[0.0, 0.0, 45.0],
                                                aug_rot_resolution=5),
                    no_skip_connection=False, no_perceiver=False, no_language=False, which_arm='right',
                    variant='two_policies', crop_target_obj_voxel=False, crop_radius=0.0,
                    randomizations_crop_point=False, arm_pred_loss=False, arm_id_to_proprio=False),
        rlbench=dict(cameras=['front', 'left_shoulder', 'right_shoulder', 'wrist'], camera_resolution=[128, 128],
                     scene_bounds=[-0.3, -0.5, 0.6, 0.7, 0.5, 1.6]),
        replay=dict(batch_size=16), ddp=dict(num_devices=1), framework=dict(training_iterations=40000, wandb_logging=None))
    for k, v in over.items():
        path = k.split('__')
        d = cfg
        for p in path[:-1]:
            d = d[p]
        d[path[-1]] = v
    return _ns(cfg)


def replay_schema(cameras, voxel_sizes, image_size=(128, 128), which_arm='right', crop_target_obj_voxel=False,
                  arm_pred_loss=False, arm_id_to_proprio=False):
    """[(name, shape, dtype)] exactly as launch_utils.create_replay declares them (:56-145)."""
    if which_arm in ('dominant', 'assistive'):
        low = LOW_DIM_DOMINANT_ASSISTIVE_SIZE + (1 if arm_id_to_proprio else 0)
    else:
        low = LOW_DIM_SIZE
    el = [('low_dim_state', (low,), np.float32)]
    for c in cameras:
        el += [('%s_rgb' % c, (3, *image_size), np.float32), ('%s_point_cloud' % c, (3, *image_size), np.float32),
               ('%s_camera_extrinsics' % c, (4, 4), np.float32), ('%s_camera_intrinsics' % c, (3, 3), np.float32)]
    el += [('trans_action_indicies', (3 * len(voxel_sizes),), np.int32), ('rot_grip_action_indicies', (4,), np.int32),
           ('ignore_collisions', (1,), np.int32), ('gripper_pose', (7,), np.float32), ('lang_goal_emb', (1024,), np.float32),
           ('lang_token_embs', (77, 512), np.float32), ('task', (), str), ('lang_goal', (1,), object)]
    if arm_pred_loss:
        el.append(('label', (1,), np.int32))
    if crop_target_obj_voxel:
        el.append(('target_object_scene_bounds', (6,), np.float32))
    return el


def create_replay(batch_size, timesteps, prioritisation, task_uniform, save_dir, cameras, voxel_sizes,
                  image_size=[128, 128], replay_size=3e5, which_arm='right', crop_target_obj_voxel=False,
                  arm_pred_loss=False, arm_id_to_proprio=False):
    if which_arm == 'both':
        raise NotImplementedError("which_arm='both' belongs to the one_policy_more_heads baseline (SURVEY.md a25)")
    try:
        from yarr.replay_buffer.replay_buffer import ReplayElement
        from yarr.replay_buffer.uniform_replay_buffer import ObservationElement
        from yarr.replay_buffer.task_uniform_replay_buffer import TaskUniformReplayBuffer
    except Exception as e:  # noqa: BLE001
        raise ImportError('create_replay needs YARR (replay store is not rebuilt here); use replay_schema() for the '
                          'element list') from e
    obs_names = {'low_dim_state', 'target_object_scene_bounds'}
    elements = []
    for name, shape, dt in replay_schema(cameras, voxel_sizes, tuple(image_size), which_arm, crop_target_obj_voxel,
                                         arm_pred_loss, arm_id_to_proprio):
        is_obs = name in obs_names or any(name.endswith(s) for s in ('_rgb', '_point_cloud', '_camera_extrinsics', '_camera_intrinsics'))
        elements.append((ObservationElement if is_obs else ReplayElement)(name, shape, dt))
    return TaskUniformReplayBuffer(save_dir=save_dir, batch_size=batch_size, timesteps=timesteps,
                                   replay_capacity=int(replay_size), action_shape=(8,), action_dtype=np.float32,
                                   reward_shape=(), reward_dtype=np.float32, update_horizon=1,
                                   observation_elements=elements,
                                   extra_replay_elements=[ReplayElement('demo', (), bool)])


def create_agent(cfg):
    """reference :663-829 (variant 'two_policies': the single-arm / acting / stabilizing policies)."""
    if cfg.method.variant == 'one_policy_more_heads':
        raise NotImplementedError('one_policy_more_heads (PerceiverVoxelLang2RobotsEncoder) is the upstream baseline, '
                                  'not the VoxAct-B method (SURVEY.md section 8a row a25)')
    depth_0bounds = cfg.rlbench.scene_bounds
    cam_resolution = cfg.rlbench.camera_resolution
    num_rotation_classes = int(360. // cfg.method.rotation_resolution)
    agents = []
    for depth, vox_size in enumerate(cfg.method.voxel_sizes):
        last = depth == len(cfg.method.voxel_sizes) - 1
        if cfg.method.which_arm in ('dominant', 'assistive'):
            low_dim_size = LOW_DIM_DOMINANT_ASSISTIVE_SIZE + (1 if cfg.method.arm_id_to_proprio else 0)
        else:
            low_dim_size = LOW_DIM_SIZE
        m = cfg.method
        enc = PerceiverVoxelLangEncoder(
            depth=m.transformer_depth, iterations=m.transformer_iterations, voxel_size=vox_size, initial_dim=3 + 3 + 1 + 3,
            low_dim_size=low_dim_size, layer=depth, num_rotation_classes=num_rotation_classes if last else 0,
            num_grip_classes=2 if last else 0, num_collision_classes=2 if last else 0, input_axis=3,
            num_latents=m.num_latents, latent_dim=m.latent_dim, cross_heads=m.cross_heads, latent_heads=m.latent_heads,
            cross_dim_head=m.cross_dim_head, latent_dim_head=m.latent_dim_head, weight_tie_layers=False,
            activation=m.activation, pos_encoding_with_lang=m.pos_encoding_with_lang, input_dropout=m.input_dropout,
            attn_dropout=m.attn_dropout, decoder_dropout=m.decoder_dropout, lang_fusion_type=m.lang_fusion_type,
            voxel_patch_size=m.voxel_patch_size, voxel_patch_stride=m.voxel_patch_stride,
            no_skip_connection=m.no_skip_connection, no_perceiver=m.no_perceiver, no_language=m.no_language,
            final_dim=m.final_dim, arm_pred_loss=m.arm_pred_loss)
        agents.append(QAttentionPerActBCAgent(
            layer=depth, coordinate_bounds=depth_0bounds, perceiver_encoder=enc, camera_names=cfg.rlbench.cameras,
            voxel_size=vox_size, bounds_offset=m.bounds_offset[depth - 1] if depth > 0 else None,
            image_crop_size=m.image_crop_size, lr=m.lr, training_iterations=cfg.framework.training_iterations,
            lr_scheduler=m.lr_scheduler, num_warmup_steps=m.num_warmup_steps, trans_loss_weight=m.trans_loss_weight,
            rot_loss_weight=m.rot_loss_weight, grip_loss_weight=m.grip_loss_weight,
            collision_loss_weight=m.collision_loss_weight, include_low_dim_state=True, image_resolution=cam_resolution,
            batch_size=cfg.replay.batch_size, voxel_feature_size=3, lambda_weight_l2=m.lambda_weight_l2,
            num_rotation_classes=num_rotation_classes, rotation_resolution=m.rotation_resolution,
            transform_augmentation=m.transform_augmentation.apply_se3,
            transform_augmentation_xyz=m.transform_augmentation.aug_xyz,
            transform_augmentation_rpy=m.transform_augmentation.aug_rpy,
            transform_augmentation_rot_resolution=m.transform_augmentation.aug_rot_resolution,
            optimizer_type=m.optimizer, num_devices=cfg.ddp.num_devices, crop_target_obj_voxel=m.crop_target_obj_voxel,
            wandb_run=cfg.framework.wandb_logging, arm_pred_loss=m.arm_pred_loss,
            randomizations_crop_point=m.randomizations_crop_point))
    rotation_agent = QAttentionStackAgent(qattention_agents=agents, rotation_resolution=cfg.method.rotation_resolution,
                                          camera_names=cfg.rlbench.cameras)
    return PreprocessAgent(pose_agent=rotation_agent)
