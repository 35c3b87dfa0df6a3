"""Data-parallel paths of molgym_amd.ppo on CPU: world_size 2 over gloo must reproduce the single-process result.

* `train`, autograd branch (any AbstractActorCritic: compute_loss + backward) -- TinyAC;
* `train`, DEVICE branch (agents with prepare_rollout / ppo_minibatch, i.e. the code the HIP agents run: loss_scale,
  device-side statistics, empty slices as zeros) -- TinyDeviceAC implements that interface with torch on the CPU;
* a remainder mini-batch smaller than the world size (one rank's slice is EMPTY);
* rollout sharding: every rank holds its own environments' buffer, `gather_rollout` standardises the advantages
  globally and all-gathers; the result must equal `get_data()` of the merged single-process buffer.
The HIP kernels themselves need a GPU; what is under test here is the sharding / reduction logic around them."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from molgym_amd import ppo
from molgym_amd.buffer import DynamicPPOBuffer, PPOBufferContainer
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


class _Rollout:
    def __init__(self, ac, data):
        self.ac, self.data = ac, data

    def minibatch(self, indices):
        return ppo.collect_data_batch(self.data, np.asarray(indices))


class TinyDeviceAC(TinyAC):
    """the interface `ppo.train` takes its device path on (CovariantAC / SchNetAC), in plain torch"""

    def prepare_rollout(self, data):
        return _Rollout(self, data)

    def ppo_minibatch(self, batch, clip_ratio, vf_coef, entropy_coef, loss_scale=1.0, slot=0):
        loss, info = ppo.compute_loss(self, batch, clip_ratio, vf_coef, entropy_coef)
        (loss * loss_scale).backward()  # accumulates into .grad like the kernels do
        return torch.tensor([info[k] for k in ppo.KEYS], dtype=torch.float64)


def _data(n):
    d = make_batch(n, 7, [0, 9, 16], seed=3)
    ac = TinyAC()
    with torch.no_grad():
        d['logp'] = ac.step(d['obs'], d['act'])['logp'].double().numpy() + 0.01
    return d


def _init(rank, world, port):
    if world > 1:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)


def _done(world):
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def _run_train(rank, world, port, out, kind, n):
    _init(rank, world, port)
    ac = TinyDeviceAC() if kind == 'device' else TinyAC()
    opt = torch.optim.Adam(ac.parameters(), lr=1e-2)
    np.random.seed(11 + 100 * rank)  # ranks do NOT share the numpy stream: rank 0's permutation is broadcast
    infos = ppo.train(ac, opt, _data(n), mini_batch_size=8, clip_ratio=0.2, target_kl=1e9, vf_coef=0.5,
                      entropy_coef=0.01, gradient_clip=0.5, max_num_steps=3)
    if rank == 0:
        torch.save({'sd': ac.state_dict(), 'infos': {k: v for k, v in infos.items() if k != 'time'}}, out)
    _done(world)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _check_train(tmp_path, kind, n):
    single, double = str(tmp_path / 'w1.pt'), str(tmp_path / 'w2.pt')
    _run_train(0, 1, 0, single, kind, n)
    mp.spawn(_run_train, args=(2, _free_port(), double, kind, n), nprocs=2, join=True)
    a, b = torch.load(single), torch.load(double)
    assert a['infos']['num_opt_steps'] == 3
    for k in a['sd']:
        assert torch.isfinite(b['sd'][k]).all()
        assert torch.allclose(a['sd'][k], b['sd'][k], atol=1e-6, rtol=1e-5), k
    for k in a['infos']:
        assert np.isfinite(b['infos'][k])
        assert abs(a['infos'][k] - b['infos'][k]) < 1e-6 * max(1.0, abs(a['infos'][k])), k


def test_world2_equals_world1(tmp_path):
    _check_train(tmp_path, 'autograd', 23)  # 23: uneven slices and a remainder mini-batch of 7


def test_world2_equals_world1_device_branch(tmp_path):
    _check_train(tmp_path, 'device', 23)


def test_empty_slice_of_a_remainder_minibatch(tmp_path):
    """17 samples, mini-batches of 8: the remainder holds ONE sample, rank 0's slice of it is empty"""
    _check_train(tmp_path, 'autograd', 17)
    _check_train(tmp_path, 'device', 17)


# ---- rollout sharding ------------------------------------------------------------------------------------------------
def _fill(container, env_ids, seed):
    """deterministic fake trajectories for the environments `env_ids` (content depends on the GLOBAL env id only)"""
    for slot, env in enumerate(env_ids):
        rng = np.random.default_rng(seed + env)
        d = make_batch(6, 7, [0, 9, 16], seed=seed + env)
        buf = container.buffers[slot]
        for t in range(6):
            terminal = t in (2, 5) if env % 2 == 0 else t == 5
            buf.store(d['obs'][t], d['act'][t], float(rng.normal()), d['obs'][(t + 1) % 6], terminal,
                      float(rng.normal()), float(rng.normal(-4, 1)))
            if terminal:
                buf.finish_path(0.0)
        if not buf.is_finished():
            buf.finish_path(float(rng.normal()))


def _run_gather(rank, world, port, out):
    _init(rank, world, port)
    envs = list(range(4))
    mine = envs[rank * 4 // world:(rank + 1) * 4 // world]
    container = PPOBufferContainer(size=len(mine), gamma=0.99, lam=0.97)
    _fill(container, mine, seed=50)
    data = ppo.gather_rollout(container.merge())
    if rank == 0:
        torch.save({k: (v if k == 'obs' else np.asarray(v)) for k, v in data.items()}, out)
    _done(world)


def test_rollout_sharding_matches_single_process(tmp_path):
    single, double = str(tmp_path / 'g1.pt'), str(tmp_path / 'g2.pt')
    _run_gather(0, 1, 0, single)
    mp.spawn(_run_gather, args=(2, _free_port(), double), nprocs=2, join=True)
    a, b = torch.load(single, weights_only=False), torch.load(double, weights_only=False)
    assert a['obs'] == b['obs'] and len(a['obs']) == 24
    for k in ('act', 'ret', 'adv', 'logp'):
        assert np.allclose(a[k], b[k], rtol=1e-12, atol=1e-12), k
    assert abs(a['adv'].mean()) < 1e-12 and abs(a['adv'].std() - 1) < 1e-12
