"""Data-parallel branch of molgym_amd.ppo.train on CPU: world_size 2 over gloo must reproduce the
single-process update (same permutation, sliced mini-batches, gradient scale B_local/B_global, one
all-reduce per epoch).  The agent here is a small differentiable stand-in with the AbstractActorCritic
contract -- the HIP agent itself needs a GPU; what is under test is the sharding/reduction logic."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from molgym_amd import ppo
from molgym_amd.synthetic import make_batch


class TinyAC(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.net = torch.nn.Sequential(torch.nn.Linear(9, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))

    def _feats(self, observations, actions):
        rows = []
        for (canvas, bag), a in zip(observations, actions):
            pos = np.array([xyz for _, xyz in canvas])
            rows.append(np.concatenate([pos.mean(0), [len(bag), sum(bag), pos.std()], a[2:5]]))
        return torch.tensor(np.array(rows), dtype=torch.float32)

    def step(self, observations, actions=None):
        out = self.net(self._feats(observations, np.asarray(actions)))
        return {'logp': -out[:, 0].abs() - 1.0, 'ent': out[:, 1].abs(), 'v': out[:, 2]}


def _data():
    d = make_batch(23, 7, [0, 9, 16], seed=3)  # 23: uneven slices and a remainder mini-batch
    ac = TinyAC()
    with torch.no_grad():
        d['logp'] = ac.step(d['obs'], d['act'])['logp'].double().numpy() + 0.01
    return d


def _run(rank, world, port, out):
    if world > 1:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    ac = TinyAC()
    opt = torch.optim.Adam(ac.parameters(), lr=1e-2)
    np.random.seed(11)
    infos = ppo.train(ac, opt, _data(), mini_batch_size=8, clip_ratio=0.2, target_kl=1e9, vf_coef=0.5,
                      entropy_coef=0.01, gradient_clip=0.5, max_num_steps=3)
    if rank == 0:
        torch.save({'sd': ac.state_dict(), 'infos': {k: v for k, v in infos.items() if k != 'time'}}, out)
    if world > 1:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_world2_equals_world1(tmp_path):
    single, double = str(tmp_path / 'w1.pt'), str(tmp_path / 'w2.pt')
    _run(0, 1, 0, single)
    mp.spawn(_run, args=(2, _free_port(), double), nprocs=2, join=True)
    a, b = torch.load(single), torch.load(double)
    for k in a['sd']:
        assert torch.allclose(a['sd'][k], b['sd'][k], atol=1e-6, rtol=1e-5), k
    for k in a['infos']:
        assert abs(a['infos'][k] - b['infos'][k]) < 1e-6 * max(1.0, abs(a['infos'][k])), k
