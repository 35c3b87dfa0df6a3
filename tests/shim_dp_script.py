"""A single-process training script in the style of /root/reference/scripts/run.py (it imports `molgym.*` only and knows
nothing about ranks), launched by tests/test_dp_gloo.py under `python -m torch.distributed.run --nproc-per-node 2`."""
import json
import os
import sys

import numpy as np
import torch

import molgym  # noqa: F401  (the shim: brings up torch.distributed from the launcher's environment)
from molgym.env_container import SimpleEnvContainer
from molgym.ppo import batch_ppo
from molgym.tools import util

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fake_env import FakeMolEnv  # noqa: E402
from test_dp_gloo import TinyDeviceAC  # noqa: E402

ZS = [0, 9, 16]


class RolloutAC(TinyDeviceAC):
    """TinyDeviceAC + the rollout side of step(): random valid actions from the torch RNG"""

    def step(self, observations, actions=None):
        if actions is not None:
            return super().step(observations, actions)
        B = len(observations)
        a = torch.zeros(B, 6)
        a[:, 1] = torch.randint(1, 3, (B, )).float()
        a[:, 2] = 1.0 + torch.rand(B)
        d = torch.randn(B, 3)
        a[:, 3:6] = d / d.norm(dim=1, keepdim=True)
        out = super().step(observations, a.numpy())
        acts = []
        for row, (canvas, bag) in zip(a.numpy(), observations):
            atoms = [xyz for label, xyz in canvas if ZS[label] != 0]
            base = np.asarray(atoms[0]) if atoms else np.zeros(3)
            acts.append((int(row[1]), tuple(base + row[2] * row[3:6])))
        return {'a': a, 'actions': acts, 'logp': out['logp'].detach(), 'ent': out['ent'].detach(), 'v': out['v'].detach()}


class Saver:
    def __init__(self):
        self.records = {}

    def save(self, obj, name):
        self.records.setdefault(name, []).append(json.loads(json.dumps(obj)))


def main():
    out = sys.argv[1]
    util.set_seeds(seed=0)  # like run.py: the same seed on every rank
    ac = RolloutAC()
    before = {k: v.clone() for k, v in ac.state_dict().items()}
    num_envs, steps = int(os.environ.get('MG_TEST_NUM_ENVS', '4')), int(os.environ.get('MG_TEST_STEPS', '16'))
    all_envs = [FakeMolEnv(5, ZS, (0, 1 + i % 2, 2)) for i in range(num_envs)]
    for i, e in enumerate(all_envs):
        e.env_id = i
    envs = SimpleEnvContainer(all_envs)
    eval_envs = SimpleEnvContainer([FakeMolEnv(5, ZS, (0, 2, 1))])
    saver = Saver()
    stored = []
    import molgym_amd.ppo as hot
    orig_rollout = hot.batch_rollout

    def counting_rollout(ac, envs, buffer_container, num_steps=None, num_episodes=None, **kw):
        res = orig_rollout(ac, envs, buffer_container, num_steps=num_steps, num_episodes=num_episodes, **kw)
        if num_steps is not None:
            stored.append(sum(b.current_index for b in buffer_container.buffers))
        return res

    hot.batch_rollout = counting_rollout
    batch_ppo(envs=envs, eval_envs=eval_envs, ac=ac, optimizer=torch.optim.Adam(ac.parameters(), lr=1e-2), gamma=1.0,
              max_num_steps=2 * steps, num_steps_per_iter=steps, mini_batch_size=4, clip_ratio=0.2, vf_coef=0.5, entropy_coef=0.01,
              max_num_train_iters=2, lam=0.97, target_kl=1e9, gradient_clip=0.5, eval_freq=1, num_eval_episodes=1,
              info_saver=saver)
    import torch.distributed as dist
    rank = dist.get_rank()
    torch.save({'world': dist.get_world_size(), 'initialised_by_shim': hot.DP_SHARD_GLOBAL_CONFIG,
                'local_envs': envs.get_size(), 'env_ids': [e.env_id for e in envs.environments], 'sd': ac.state_dict(),
                'moved': any(not torch.equal(before[k], v) for k, v in ac.state_dict().items()),
                'rollout_seed_probe': float(np.random.rand()) + float(torch.rand(1)),
                'train_log': saver.records.get('train', []), 'steps_stored_per_iteration': stored[0]},
               f'{out}.rank{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
