"""PreprocessAgent -- drop-in for peract/helpers/preprocess_agent.py:9-126 (drops the T=1 axis, rgb -> [-1,1], float cast)."""
from typing import List

import torch

from ..yarr_agent import Agent, Summary, ActResult, ScalarSummary, HistogramSummary


class PreprocessAgent(Agent):

    def __init__(self, pose_agent: Agent, norm_rgb: bool = True):
        self._pose_agent = pose_agent
        self._norm_rgb = norm_rgb

    def build(self, training: bool, device: torch.device = None):
        self._pose_agent.build(training, device)

    def _norm_rgb_(self, x):
        return (x.float() / 255.0) * 2.0 - 1.0

    def _prep(self, d):
        return {k: (self._norm_rgb_(v) if (self._norm_rgb and 'rgb' in k) else v.float()) for k, v in d.items()}

    def update(self, step: int, replay_sample: dict) -> dict:
        replay_sample = {k: v[:, 0] if len(v.shape) > 2 else v for k, v in replay_sample.items()}   # :25
        replay_sample = self._prep(replay_sample)
        self._replay_sample = replay_sample
        return self._pose_agent.update(step, replay_sample)

    def act(self, step: int, observation: dict, deterministic=False, which_arm=None, new_scene_bounds=None,
            dominant_assitive_policy=False, ep_number=0, is_real_robot=False) -> ActResult:
        observation.update(self._prep(observation))
        act_res = self._pose_agent.act(step, observation, deterministic, which_arm, new_scene_bounds,
                                       dominant_assitive_policy, ep_number, is_real_robot)
        if is_real_robot:
            return act_res
        act_res.replay_elements.update({'demo': False})
        return act_res

    def update_summaries(self) -> List[Summary]:
        prefix = 'inputs'
        sums = []
        if hasattr(self, '_replay_sample') and 'demo' in self._replay_sample:
            sums.append(ScalarSummary('%s/demo_proportion' % prefix, self._replay_sample['demo'].float().mean()))
        if hasattr(self, '_replay_sample') and 'low_dim_state' in self._replay_sample:
            sums.append(HistogramSummary('%s/low_dim_state' % prefix, self._replay_sample['low_dim_state']))
        local, wandb_dict = self._pose_agent.update_summaries()
        sums.extend(local)
        return sums, wandb_dict

    def act_summaries(self) -> List[Summary]:
        return self._pose_agent.act_summaries()

    def load_weights(self, savedir: str):
        self._pose_agent.load_weights(savedir)

    def load_weight(self, ckpt_file: str):
        self._pose_agent.load_weight(ckpt_file)

    def save_weights(self, savedir: str):
        self._pose_agent.save_weights(savedir)

    def reset(self) -> None:
        self._pose_agent.reset()
