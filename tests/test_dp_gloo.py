"""Data-parallel paths of molgym_amd.ppo on CPU: world_size 2 over gloo must reproduce the single-process result.

* `train`, autograd branch (any AbstractActorCritic: compute_loss + backward) -- TinyAC;
* `train`, DEVICE branch (agents with prepare_rollout / ppo_minibatch, i.e. the code the HIP agents run: loss_scale,
  device-side statistics, empty slices as zeros) -- TinyDeviceAC implements that interface with torch on the CPU;
* WHOLE mini-batches dealt round-robin to the ranks (32 samples = 4 mini-batches of 8: no slicing at all), mixed with a
  sliced leftover (23 samples: two whole + a remainder of 7), and the partition itself (`shard_epoch`);
* a remainder mini-batch smaller than the world size (one rank's slice is EMPTY);
* rollout sharding: every rank holds its own environments' buffer, `gather_rollout` all-gathers ONE float64 matrix per rank
  (observations as arrays) and standardises the advantages over the merged buffer; the result must equal `get_data()` of
  the merged single-process buffer;
* the `molgym` shim initialising the process group for an UNCHANGED single-process script under torch.distributed.run.
The HIP kernels themselves need a GPU; what is under test here is the sharding / reduction logic around them."""
import os
import socket

import numpy as np
import pytest
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


class TinyRunaheadAC(TinyDeviceAC):
    """everything `ppo._train_runahead` calls on the HIP agents (FlatThetaAgent: statistics accumulated on the device, the
    epoch's KL test / norm / clip as `ppo_epoch_end` with a latching stop flag, the Adam step gated by that flag and taken
    back by `adam_unstep`), in plain torch on the CPU: what is under test is the loop around them"""
    flat_gradient_on_host = True

    def ppo_minibatch(self, batch, clip_ratio, vf_coef, entropy_coef, loss_scale=1.0, slot=0, stats_accum=None):
        stats = super().ppo_minibatch(batch, clip_ratio, vf_coef, entropy_coef, loss_scale, slot)
        stats_accum += stats * loss_scale
        return None

    def _grads(self):
        return [p.grad for p in self.parameters()]

    def grad_norm_clip(self, max_norm=0.0):
        norm = torch.norm(torch.stack([torch.norm(g, 2) for g in self._grads()]), 2)
        if max_norm > 0:
            coef = max_norm / (norm + 1e-6)
            if coef < 1:
                for g in self._grads():
                    g.mul_(coef)
        return norm.reshape(1)

    def ppo_epoch_end(self, max_norm, stats_accum, num_minibatches, kl_limit, rec, stop_flag):
        stats = stats_accum / max(int(num_minibatches), 1)
        stop = bool(stop_flag.item()) or stats[4].item() > kl_limit
        norm = self.grad_norm_clip(0.0 if stop else max_norm)
        rec[:6] = stats
        rec[6] = norm.double()
        rec[7] = 1.0 if stop else 0.0
        if stop:
            stop_flag.fill_(1)

    def adam_supported(self, optimizer):
        return True

    def adam_step(self, optimizer, skip_flag=None):
        self.steps_issued = getattr(self, 'steps_issued', 0) + 1
        if skip_flag is None or skip_flag.item() == 0:
            optimizer.step()
            self.steps_taken = getattr(self, 'steps_taken', 0) + 1
        return True

    def adam_unstep(self, optimizer, count):
        self.unstepped = getattr(self, 'unstepped', 0) + count


def _kl_limited_data(n):
    d = _data(n)
    d['logp'] = d['logp'] + 0.05  # approx_kl = mean(old - new) starts at 0.06 and grows with the updates
    return d


def _run_runahead(rank, world, port, out, runahead):
    os.environ['MOLGYM_RUNAHEAD'] = '1' if runahead else '0'
    _init(rank, world, port)
    ac = TinyRunaheadAC()
    opt = torch.optim.Adam(ac.parameters(), lr=3e-2)
    np.random.seed(11 + 100 * rank)
    infos = ppo.train(ac, opt, _kl_limited_data(32), mini_batch_size=8, clip_ratio=0.2, target_kl=0.046, vf_coef=0.5,
                      entropy_coef=0.01, gradient_clip=0.5, max_num_steps=6)
    if rank == 0:
        torch.save({'sd': ac.state_dict(), 'infos': {k: v for k, v in infos.items() if k != 'time'},
                    'rng': np.random.get_state()[1], 'rng_pos': np.random.get_state()[2],
                    'issued': getattr(ac, 'steps_issued', 0), 'taken': getattr(ac, 'steps_taken', 0),
                    'unstepped': getattr(ac, 'unstepped', 0)}, out)
    _done(world)


def test_runahead_loop_with_early_kl_stop_world2_equals_the_synchronous_loop(tmp_path):
    """`_train_runahead` (the epochs without a host round trip: KL test on the device, gated Adam, epoch i + 1 issued before epoch
    i's record is read) at world 2 over gloo against the reference-shaped synchronous loop in one process: same theta, same
    statistics, same number of optimizer steps, and the numpy RNG left where the synchronous loop leaves it (the speculative
    epoch's permutation is taken back)."""
    sync, ahead = str(tmp_path / 'sync.pt'), str(tmp_path / 'ahead.pt')
    _run_runahead(0, 1, 0, sync, False)
    mp.spawn(_run_runahead, args=(2, _free_port(), ahead, True), nprocs=2, join=True)
    os.environ.pop('MOLGYM_RUNAHEAD', None)
    a, b = torch.load(sync, weights_only=False), torch.load(ahead, weights_only=False)
    assert 1 <= a['infos']['num_opt_steps'] < 5, a['infos']  # the KL limit stops the loop early, after at least one update
    assert b['infos']['num_opt_steps'] == a['infos']['num_opt_steps']
    assert b['taken'] == a['infos']['num_opt_steps'] and b['issued'] - b['unstepped'] == b['taken']
    assert b['issued'] > b['taken']  # the run-ahead loop did issue an epoch behind the one that stopped
    for k in a['sd']:
        assert torch.allclose(a['sd'][k], b['sd'][k], atol=1e-6, rtol=1e-5), k
    for k in a['infos']:
        assert abs(a['infos'][k] - b['infos'][k]) < 1e-6 * max(1.0, abs(a['infos'][k])), k
    assert b['rng_pos'] == a['rng_pos'] and np.array_equal(a['rng'], b['rng'])


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


def test_shard_epoch_covers_every_minibatch_exactly_once():
    rng = np.random.default_rng(0)
    for n, mb, world in ((140, 140, 8), (1400, 140, 8), (23, 8, 2), (17, 8, 2), (64, 8, 3), (5, 8, 4)):
        batches = list(ppo.get_batch_generator(np.arange(n), mb))
        seen, shares = [], [0.0] * len(batches)
        for rank in range(world):
            work = ppo.shard_epoch(batches, rank, world)
            whole = [w for w in work if w[1] == 1.0 and len(w[0]) == mb]
            if len(batches) >= world:  # B_local stays at the mini-batch size wherever a whole one is available
                assert len(whole) >= len(batches) // world
            for idx, share in work:
                seen += list(idx)
                k = [i for i, b in enumerate(batches) if len(idx) == 0 or idx[0] in b]
                if len(idx):
                    shares[k[0]] += share
                    assert abs(share - len(idx) / len(batches[k[0]])) < 1e-12
        assert sorted(seen) == list(range(n)) and all(abs(s - 1.0) < 1e-12 for s in shares)
    del rng


def test_world2_equals_world1(tmp_path):
    _check_train(tmp_path, 'autograd', 23)  # 23: two whole mini-batches (one per rank) and a sliced remainder of 7


def test_whole_minibatch_sharding_world2_equals_world1(tmp_path):
    """32 samples = 4 mini-batches of 8: every rank evaluates two WHOLE mini-batches, nothing is sliced"""
    _check_train(tmp_path, 'autograd', 32)
    _check_train(tmp_path, 'device', 32)


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
    from molgym_amd.observations import ParsedObservations
    assert isinstance(b['obs'], ParsedObservations) and len(a['obs']) == len(b['obs']) == 24
    assert np.array_equal(ParsedObservations.from_list(a['obs']).to_matrix(), b['obs'].to_matrix())  # bit for bit
    assert b['obs'][5] == a['obs'][5]  # ... and an element reads back as the reference's tuple
    for k in ('act', 'ret', 'adv', 'logp'):
        assert np.allclose(a[k], b[k], rtol=1e-12, atol=1e-12), k
    assert abs(a['adv'].mean()) < 1e-12 and abs(a['adv'].std() - 1) < 1e-12


# ---- an unchanged single-process script under torch.distributed.run ---------------------------------------------------
def test_shim_initialises_the_process_group_for_an_unchanged_script(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 script.py` where the script only imports `molgym` and calls
    `molgym.ppo.batch_ppo` like scripts/run.py does (it knows nothing about ranks): the shim brings up the process group
    (gloo here), every rank keeps half of the 4 environments and half of the steps, the rollout RNG streams differ by
    rank, the replicas stay identical, and rank 0's log covers the rollout of both ranks."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / 'shim')
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), MOLGYM_DIST_BACKEND='gloo',
               OMP_NUM_THREADS='1')
    env.pop('MOLGYM_REFERENCE', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'tests', 'shim_dp_script.py'), out]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    r0, r1 = torch.load(out + '.rank0.pt', weights_only=False), torch.load(out + '.rank1.pt', weights_only=False)
    assert r0['world'] == r1['world'] == 2 and r0['initialised_by_shim'] and r1['initialised_by_shim']
    assert r0['local_envs'] == r1['local_envs'] == 2                       # 4 environments in the script, 2 per rank
    assert all(torch.equal(r0['sd'][k], r1['sd'][k]) for k in r0['sd'])     # identical replicas after the updates
    assert r0['moved']
    assert r0['rollout_seed_probe'] != r1['rollout_seed_probe']             # decorrelated rollouts
    # rank 0 logged 16 steps per iteration = the GLOBAL num_steps_per_iter, episode statistics over both ranks
    assert r0['train_log'] and all(rec['total_num_steps'] % 16 == 0 for rec in r0['train_log'])
    assert r0['steps_stored_per_iteration'] == r1['steps_stored_per_iteration'] == 8


def test_shim_deals_an_odd_number_of_environments_to_four_ranks(tmp_path):
    """`torchrun --nproc-per-node 4 script.py` with num_envs = 5 (not a multiple of the world size): every environment is
    stepped by exactly one rank (shares 1, 1, 1, 2), every environment takes the same number of steps per iteration, the
    job as a whole stores the script's GLOBAL num_steps_per_iter, and the replicas stay identical."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / 'shim4')
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), MOLGYM_DIST_BACKEND='gloo',
               OMP_NUM_THREADS='1', MG_TEST_NUM_ENVS='5', MG_TEST_STEPS='20')
    env.pop('MOLGYM_REFERENCE', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '4', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'tests', 'shim_dp_script.py'), out]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    recs = [torch.load(f'{out}.rank{r}.pt', weights_only=False) for r in range(4)]
    assert all(r['world'] == 4 and r['initialised_by_shim'] for r in recs)
    ids = sorted(i for r in recs for i in r['env_ids'])
    assert ids == [0, 1, 2, 3, 4]                                            # every environment exactly once
    assert sorted(r['local_envs'] for r in recs) == [1, 1, 1, 2]
    # 20 global steps per iteration = 4 per environment: a rank stores 4 x (its environments)
    assert [r['steps_stored_per_iteration'] for r in recs] == [4 * r['local_envs'] for r in recs]
    assert sum(r['steps_stored_per_iteration'] for r in recs) == 20
    assert all(torch.equal(recs[0]['sd'][k], r['sd'][k]) for r in recs[1:] for k in recs[0]['sd'])
    assert recs[0]['moved']
    assert recs[0]['train_log'] and all(rec['total_num_steps'] % 20 == 0 for rec in recs[0]['train_log'])


def test_global_config_sharding_is_idempotent_per_container():
    """a second batch_ppo call on the same container (curriculum, resume) must not shard the already-sharded environments
    again nor reseed the RNG streams; only the (again global) step count is converted"""
    from molgym_amd import ppo

    class Box:
        def __init__(self, n):
            self.environments = list(range(n))

        def get_size(self):
            return len(self.environments)

    np.random.seed(3)
    box = Box(5)
    envs, steps = ppo._shard_global_config(box, 20, rank=3, world=4)
    assert envs.environments == [3, 4] and steps == 8
    probe = np.random.get_state()[1][:4].copy()
    envs, steps = ppo._shard_global_config(box, 40, rank=3, world=4)
    assert envs.environments == [3, 4] and steps == 16
    assert np.array_equal(np.random.get_state()[1][:4], probe)  # no second reseed
    with pytest.raises(RuntimeError):
        ppo._shard_global_config(Box(3), 12, rank=0, world=4)   # fewer environments than ranks
    with pytest.raises(RuntimeError):
        ppo._shard_global_config(Box(5), 12, rank=0, world=4)   # steps not a multiple of num_envs
