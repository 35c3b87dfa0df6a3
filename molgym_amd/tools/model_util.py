"""Model factory (mirror of /root/reference/molgym/tools/model_util.py:15-41): same `config` keys, returns the
HIP-backed agents."""
import os

import torch

from molgym_amd.agents.base import AbstractActorCritic


def build_model(config: dict, observation_space, action_space, device: torch.device) -> AbstractActorCritic:
    if config['model'] == 'internal':
        from molgym_amd.agents.internal import SchNetAC
        return SchNetAC(
            observation_space=observation_space,
            action_space=action_space,
            min_max_distance=(config['min_mean_distance'], config['max_mean_distance']),
            network_width=config['network_width'],
            device=device,
        )
    elif config['model'] == 'covariant':
        from molgym_amd.agents.covariant import CovariantAC
        return CovariantAC(
            observation_space=observation_space,
            action_space=action_space,
            min_max_distance=(config['min_mean_distance'], config['max_mean_distance']),
            network_width=config['network_width'],
            maxl=config['maxl'],
            num_cg_levels=config['num_cg_levels'],
            num_channels_hidden=config['num_channels_hidden'],
            num_channels_per_element=config['num_channels_per_element'],
            num_gaussians=config['num_gaussians'],
            bag_scale=config['bag_scale'],
            beta=float(config['beta']) if config['beta'] is not None else config['beta'],
            device=device,
        )
    raise RuntimeError(f'Model \'{config["model"]}\' is not available.')


class ModelIO:
    """Checkpoint files of the training scripts (tools/model_util.py:51-117): `<tag>_steps-<n>.model`, written with
    torch.save of the WHOLE module (the HIP agents pickle their flat parameter vector and drop device caches), the
    previous file removed unless `keep`; `load` / `load_latest` return (module, num_steps)."""
    _STEPS, _SUFFIX = '_steps-', '.model'

    def __init__(self, directory: str, tag: str, keep: bool = False) -> None:
        self.directory, self.tag, self.keep = directory, tag, keep
        self.old_path = None

    def _parse(self, path: str):
        import re
        m = re.match(rf'(?P<tag>.+){self._STEPS}(?P<num_steps>\d+){re.escape(self._SUFFIX)}', os.path.basename(path))
        return None if not m else (path, m.group('tag'), int(m.group('num_steps')))

    def save(self, module: AbstractActorCritic, num_steps: int) -> None:
        if not self.keep and self.old_path:
            os.remove(self.old_path)
        path = os.path.join(self.directory, f'{self.tag}{self._STEPS}{num_steps}{self._SUFFIX}')
        torch.save(obj=module, f=path)
        self.old_path = path

    def load(self, device: torch.device, path: str):
        info = self._parse(path)
        if info is None:
            raise RuntimeError(f"Cannot find model '{path}'")
        return torch.load(f=info[0], map_location=device, weights_only=False), info[2]

    def load_latest(self, device: torch.device):
        paths = [os.path.join(self.directory, f) for f in os.listdir(self.directory)]
        infos = [self._parse(p) for p in paths if os.path.isfile(p)]
        infos = [i for i in infos if i and i[1] == self.tag]
        if not infos:
            raise RuntimeError(f"Cannot find model to load in '{self.directory}'")
        latest = max(infos, key=lambda i: i[2])
        return torch.load(f=latest[0], map_location=device, weights_only=False), latest[2]
