"""Actor-critic interface (mirror of /root/reference/molgym/agents/base.py:10-19)."""
import abc
from typing import List, Optional

import numpy as np
import torch


class AbstractActorCritic(torch.nn.Module, abc.ABC):
    def __init__(self, observation_space, action_space):
        super().__init__()
        self.observation_space = observation_space
        self.action_space = action_space

    @abc.abstractmethod
    def step(self, observations: List, actions: Optional[np.ndarray] = None) -> dict:
        raise NotImplementedError
