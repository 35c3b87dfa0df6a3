"""The helpers of /root/reference/molgym/tools/util.py the PPO loop itself touches (formula / logging / argparse
helpers stay with the reference: they are not on the hot path).  Same names and behaviour:
to_numpy :46-47, count_vars :56-57, compute_gradient_norm :61-69, discount_cumsum :72-87, set_seeds :90-92,
RolloutSaver :157-168 (pickles of DynamicPPOBuffer objects), InfoSaver :171-183 (JSON lines), init_device :186-194,
get_optimizer :197-205."""
import json
import logging
import os
import pickle
from typing import Iterable

import numpy as np
import torch

from ..buffer import discount_cumsum  # noqa: F401  (re-exported under its reference name)
from ..ppo import compute_gradient_norm, to_numpy  # noqa: F401


def count_vars(module: torch.nn.Module) -> int:
    return int(sum(np.prod(p.shape) for p in module.parameters()))


def set_seeds(seed: int) -> None:
    np.random.seed(seed)
    torch.manual_seed(seed)


class RolloutSaver:
    def __init__(self, directory: str, tag: str):
        self.directory, self.tag = directory, tag

    def save(self, obj: object, num_steps: int, info: str):
        path = os.path.join(self.directory, f'{self.tag}_steps-{num_steps}_{info}.pkl')
        logging.debug(f'Saving rollout: {path}')
        with open(path, mode='wb') as f:
            pickle.dump(obj, f)


class InfoSaver:
    def __init__(self, directory: str, tag: str):
        self.directory, self.tag = directory, tag

    def save(self, obj: object, name: str):
        path = os.path.join(self.directory, f'{self.tag}_{name}.txt')
        logging.debug(f'Saving info: {path}')
        with open(path, mode='a') as f:
            f.write(json.dumps(obj) + '\n')


def init_device(device_str: str) -> torch.device:
    if device_str == 'cuda':
        assert torch.cuda.is_available(), 'No CUDA device available!'
        logging.info('CUDA Device: {}'.format(torch.cuda.current_device()))
        torch.cuda.init()
        return torch.device('cuda')
    logging.info('Using CPU')
    return torch.device('cpu')


def get_optimizer(name: str, learning_rate: float, parameters: Iterable[torch.Tensor]):
    if name not in ('adam', 'amsgrad'):
        raise RuntimeError(f"Unknown optimizer '{name}'")
    return torch.optim.Adam(parameters, lr=learning_rate, amsgrad=(name == 'amsgrad'))
