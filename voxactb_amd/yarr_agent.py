"""Boundary types restated from YARR (reference: YARR/yarr/agents/agent.py:5-78) so that the agent stack can be
imported on a box without YARR.  If `yarr` is importable its own classes are used, which makes the agents below
real subclasses of the caller's `yarr.agents.agent.Agent`."""
from abc import ABC, abstractmethod
from typing import Any, List

try:  # pragma: no cover - depends on the host environment
    from yarr.agents.agent import (Agent, ActResult, Summary, ScalarSummary, HistogramSummary, ImageSummary,
                                   TextSummary, VideoSummary)
except Exception:  # noqa: BLE001
    class Summary(object):
        def __init__(self, name: str, value: Any):
            self.name = name
            self.value = value

    class ScalarSummary(Summary):
        pass

    class HistogramSummary(Summary):
        pass

    class ImageSummary(Summary):
        pass

    class TextSummary(Summary):
        pass

    class VideoSummary(Summary):
        def __init__(self, name: str, value: Any, fps: int = 30):
            super(VideoSummary, self).__init__(name, value)
            self.fps = fps

    class ActResult(object):
        def __init__(self, action: Any, observation_elements: dict = None, replay_elements: dict = None,
                     info: dict = None):
            self.action = action
            self.observation_elements = observation_elements or {}
            self.replay_elements = replay_elements or {}
            self.info = info or {}

    class Agent(ABC):
        @abstractmethod
        def build(self, training: bool, device=None) -> None:
            pass

        @abstractmethod
        def update(self, step: int, replay_sample: dict) -> dict:
            pass

        @abstractmethod
        def act(self, step: int, observation: dict, deterministic: bool) -> ActResult:
            pass

        def reset(self) -> None:
            pass

        @abstractmethod
        def update_summaries(self) -> List[Summary]:
            pass

        @abstractmethod
        def act_summaries(self) -> List[Summary]:
            pass

        @abstractmethod
        def load_weights(self, savedir: str) -> None:
            pass

        @abstractmethod
        def save_weights(self, savedir: str) -> None:
            pass
