"""Model factory (mirror of /root/reference/molgym/tools/model_util.py:15-41): same `config` keys, returns the
HIP-backed agents."""
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
