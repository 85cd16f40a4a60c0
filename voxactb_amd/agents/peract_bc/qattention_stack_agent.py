"""QAttentionStackAgent -- drop-in for peract/agents/peract_bc/qattention_stack_agent.py:14-124: loops over the
per-depth Q-attention agents (one depth for PerAct) and turns the discrete action into the continuous 9-vector."""
from typing import List

import numpy as np
import torch

from ...helpers.rotation import discrete_euler_to_quaternion, point_to_pixel_index
from ...yarr_agent import Agent, ActResult, Summary

NAME = 'QAttentionStackAgent'


class QAttentionStackAgent(Agent):

    def __init__(self, qattention_agents, rotation_resolution: float, camera_names: List[str],
                 rotation_prediction_depth: int = 0):
        super(QAttentionStackAgent, self).__init__()
        self._qattention_agents = qattention_agents
        self._rotation_resolution = rotation_resolution
        self._camera_names = camera_names
        self._rotation_prediction_depth = rotation_prediction_depth

    def build(self, training: bool, device=None) -> None:
        self._device = device
        for qa in self._qattention_agents:
            qa.build(training, device)

    def _pixel_device(self):
        d = self._device
        return torch.device('cuda:%d' % d) if isinstance(d, int) else (torch.device('cpu') if d is None else torch.device(d))

    def update(self, step: int, replay_sample: dict) -> dict:
        total_losses = 0.
        for qa in self._qattention_agents:
            update_dict = qa.update(step, replay_sample)
            replay_sample.update(update_dict)
            total_losses += update_dict['total_loss']
        return {'total_losses': total_losses}

    def act(self, step: int, observation: dict, deterministic=False, which_arm=None, new_scene_bounds=None,
            dominant_assitive_policy=False, ep_number=0, is_real_robot=False) -> ActResult:
        observation_elements, infos = {}, {}
        trans, rot_grip, coll = [], [], []
        for depth, qagent in enumerate(self._qattention_agents):
            act_results = qagent.act(step, observation, deterministic, which_arm, new_scene_bounds,
                                     dominant_assitive_policy, ep_number, is_real_robot)
            attention_coordinate = act_results.observation_elements['attention_coordinate'].cpu().numpy()
            observation_elements['attention_coordinate_layer_%d' % depth] = attention_coordinate[0]
            t, r, c = act_results.action
            trans.append(t)
            if r is not None:
                rot_grip.append(r)
            if c is not None:
                coll.append(c)
            observation['attention_coordinate'] = act_results.observation_elements['attention_coordinate']
            observation['prev_layer_voxel_grid'] = act_results.observation_elements['prev_layer_voxel_grid']
            observation['prev_layer_bounds'] = act_results.observation_elements['prev_layer_bounds']
            if not is_real_robot:          # where the chosen voxel projects into every camera (stack agent :62-70)
                for n in self._camera_names:
                    px, py = point_to_pixel_index(attention_coordinate[0],
                                                  observation['%s_camera_extrinsics' % n][0, 0].cpu().numpy(),
                                                  observation['%s_camera_intrinsics' % n][0, 0].cpu().numpy())
                    observation['%s_pixel_coord' % n] = torch.tensor([[[py, px]]], dtype=torch.float32, device=self._pixel_device())
                    observation_elements['%s_pixel_coord' % n] = [py, px]
            infos.update(act_results.info)
        rgai = torch.cat(rot_grip, 1)[0].cpu().numpy()
        ignore_collisions = float(torch.cat(coll, 1)[0, 0].cpu().numpy())
        observation_elements['trans_action_indicies'] = torch.cat(trans, 1)[0].cpu().numpy()
        observation_elements['rot_grip_action_indicies'] = rgai
        quat = discrete_euler_to_quaternion(rgai[-4:-1], self._rotation_resolution)
        coord = act_results.observation_elements['attention_coordinate'].cpu().numpy()[0]
        if is_real_robot:
            return coord, quat, rgai[-1:]
        continuous_action = np.concatenate([coord, quat, rgai[-1:], [ignore_collisions]])
        return ActResult(continuous_action, observation_elements=observation_elements, info=infos)

    def update_summaries(self) -> List[Summary]:
        summaries, wandb_dict = [], {}
        for qa in self._qattention_agents:
            s, w = qa.update_summaries()
            summaries.extend(s)
            wandb_dict = {**wandb_dict, **w}
        return summaries, wandb_dict

    def act_summaries(self) -> List[Summary]:
        s = []
        for qa in self._qattention_agents:
            s.extend(qa.act_summaries())
        return s

    def load_weights(self, savedir: str):
        for qa in self._qattention_agents:
            qa.load_weights(savedir)

    def load_weight(self, ckpt_file: str):
        for qa in self._qattention_agents:
            qa.load_weight(ckpt_file)

    def save_weights(self, savedir: str):
        for qa in self._qattention_agents:
            qa.save_weights(savedir)


class QAttentionStackAgent2Robots(QAttentionStackAgent):
    """reference qattention_stack_agent.py:127-276: the per-depth agents return both arms' actions; `which_arm` picks the one
    that is turned into the continuous action."""

    def act(self, step: int, observation: dict, deterministic=False, which_arm=None, new_scene_bounds=None, *_ignored,
            **_ignored_kw) -> ActResult:
        # (upstream's signature ends at new_scene_bounds, :153-154; PreprocessAgent passes the single-arm extras positionally)
        if which_arm not in ('right', 'left'):
            raise NotImplementedError
        key = 'attention_coordinate_' + which_arm
        observation_elements, infos = {}, {}
        trans, rot_grip, coll = [], [], []
        for depth, qagent in enumerate(self._qattention_agents):
            act_results = qagent.act(step, observation, deterministic)
            attention_coordinate = act_results.observation_elements[key].cpu().numpy()
            observation_elements['attention_coordinate_layer_%d' % depth] = attention_coordinate[0]
            a = act_results.action
            t, r, c = (a[0], a[1], a[2]) if which_arm == 'right' else (a[3], a[4], a[5])
            trans.append(t)
            if r is not None:
                rot_grip.append(r)
            if c is not None:
                coll.append(c)
            observation['attention_coordinate'] = act_results.observation_elements[key]
            observation['prev_layer_voxel_grid'] = act_results.observation_elements['prev_layer_voxel_grid']
            observation['prev_layer_bounds'] = act_results.observation_elements['prev_layer_bounds']
            for n in self._camera_names:
                px, py = point_to_pixel_index(attention_coordinate[0],
                                              observation['%s_camera_extrinsics' % n][0, 0].cpu().numpy(),
                                              observation['%s_camera_intrinsics' % n][0, 0].cpu().numpy())
                observation['%s_pixel_coord' % n] = torch.tensor([[[py, px]]], dtype=torch.float32, device=self._pixel_device())
                observation_elements['%s_pixel_coord' % n] = [py, px]
            infos.update(act_results.info)
        rgai = torch.cat(rot_grip, 1)[0].cpu().numpy()
        ignore_collisions = float(torch.cat(coll, 1)[0, 0].cpu().numpy())
        observation_elements['trans_action_indicies'] = torch.cat(trans, 1)[0].cpu().numpy()
        observation_elements['rot_grip_action_indicies'] = rgai
        continuous_action = np.concatenate([act_results.observation_elements[key].cpu().numpy()[0],
                                            discrete_euler_to_quaternion(rgai[-4:-1], self._rotation_resolution), rgai[-1:],
                                            [ignore_collisions]])
        return ActResult(continuous_action, observation_elements=observation_elements, info=infos)
