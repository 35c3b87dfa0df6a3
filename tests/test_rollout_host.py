"""Host side of the rollout: buffers (vs the reference-generated golden GAE vectors), the process-pool environment
container (same trajectories as the serial container; environment steps run in parallel and overlap the policy
evaluation of the other environment group)."""
import os
import time

import numpy as np
import pytest

from molgym_amd import ppo
from molgym_amd.buffer import DynamicPPOBuffer, PPOBufferContainer, discount_cumsum
from molgym_amd.env_container import AsyncEnvContainer, SimpleEnvContainer
from tests.fake_env import FakeAC, FakeMolEnv

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ZS = [0, 9, 16]


def test_discount_cumsum_known_answer():
    """tests/test_tools.py:19-26 of the reference"""
    assert np.allclose(discount_cumsum(np.array([1.0, 1.0, 1.0]), 0.5), [1.75, 1.5, 1.0])


def test_buffer_matches_reference_golden():
    """G2: DynamicPPOBuffer.finish_path / get_data outputs of the reference itself (oracle/make_golden.py)"""
    g = np.load(os.path.join(GOLDEN, 'g2_gae.npz'))
    for c in (0, 1):
        rew, val, off, last = g[f'c{c}_rew'], g[f'c{c}_val'], g[f'c{c}_off'], g[f'c{c}_last']
        buf = DynamicPPOBuffer(gamma=float(g[f'c{c}_gamma']), lam=float(g[f'c{c}_lam']))
        for p in range(len(off) - 1):
            for t in range(int(off[p]), int(off[p + 1])):
                buf.store(obs=None, act=np.zeros(6), reward=float(rew[t]), next_obs=None, terminal=False,
                          value=float(val[t]), logp=0.0)
            episodic_return, length = buf.finish_path(float(last[p]))
            assert length == off[p + 1] - off[p] and episodic_return == buf.ret_buf[int(off[p])]
        assert buf.finish_path(0.0) == (None, 0)
        np.testing.assert_allclose(buf.adv_buf, g[f'c{c}_adv'], rtol=1e-12)
        np.testing.assert_allclose(buf.ret_buf, g[f'c{c}_ret'], rtol=1e-12)
        np.testing.assert_allclose(buf.get_data()['adv'], g[f'c{c}_adv_norm'], rtol=1e-12)
        assert [(lo, hi) for lo, hi, _ in buf.paths] == list(zip(off[:-1], off[1:]))


def _envs(n, work=0.0):
    return [FakeMolEnv(5, ZS, (0, 1 + i % 2, 2), work_seconds=work) for i in range(n)]


def _rollout(container_envs, steps, pipeline):
    ac = FakeAC(ZS)
    cont = PPOBufferContainer(size=container_envs.get_size(), gamma=0.99, lam=0.97)
    info = ppo.batch_rollout(ac, container_envs, cont, num_steps=steps, pipeline=pipeline)
    return cont.merge(), info


def test_async_container_reproduces_serial_rollout():
    serial, _ = _rollout(SimpleEnvContainer(_envs(6)), 6 * 7, pipeline=1)
    pool = AsyncEnvContainer(_envs(6), num_workers=4)
    try:
        assert sorted(sum(pool.groups(2), [])) == list(range(6))
        whole, _ = _rollout(pool, 6 * 7, pipeline=1)       # all environments per request
        piped, info = _rollout(pool, 6 * 7, pipeline=2)    # two groups, software pipelined
    finally:
        pool.close()
    for other in (whole, piped):
        for field in DynamicPPOBuffer.BUFFER_FIELDS:
            a, b = getattr(serial, field), getattr(other, field)
            if field in ('obs_buf', 'next_obs_buf'):
                assert a == b, field
            else:
                assert np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)), field
        assert serial.paths == other.paths
    assert info['episode_length_mean'] > 0


def test_async_container_surfaces_worker_errors():
    class Broken(FakeMolEnv):
        def step(self, action):
            raise ValueError('sparrow blew up')

    pool = AsyncEnvContainer([Broken(5, ZS, (0, 1, 1))], num_workers=1)
    try:
        pool.reset()
        with pytest.raises(RuntimeError, match='sparrow blew up'):
            pool.step([(1, (0.0, 0.0, 0.0))])
    finally:
        pool.close()


def _overlap(a, b):
    return min(a[1], b[1]) - max(a[0], b[0])


@pytest.mark.skipif((os.cpu_count() or 1) < 2, reason='needs two host cores')
def test_environment_steps_run_in_parallel_and_overlap_the_policy():
    """Structure, not wall time (which would be flaky on a loaded box): with worker processes the reward computations
    of ONE vectorised step overlap in time, and with two pipelined groups a policy evaluation (a device-side wait on
    the host) overlaps the environment step of the other group.  The serial container shows neither."""
    n, work, gpu = 8, 0.02, 0.03

    class Recorder:  # wraps a container, keeps the (start, end) spans the environments report per step
        def __init__(self, envs):
            self.envs, self.steps = envs, []

        def __getattr__(self, name):
            return getattr(self.envs, name)

        def step(self, actions):
            out = self.envs.step(actions)
            self.steps.append([i['span'] for i in out[3] if 'span' in i])
            return out

        def step_wait(self, ticket=None):
            out = self.envs.step_wait(ticket)
            self.steps.append([i['span'] for i in out[3] if 'span' in i])
            return out

    def run(envs, pipeline):
        ac = FakeAC(ZS, gpu_seconds=gpu)
        rec = Recorder(envs)
        ppo.batch_rollout(ac, rec, PPOBufferContainer(size=n, gamma=0.99, lam=0.97), num_steps=n * 4, pipeline=pipeline)
        return ac.spans, rec.steps

    def parallel_pairs(steps):
        return sum(1 for spans in steps for i in range(len(spans)) for j in range(i) if _overlap(spans[i], spans[j]) > 0.005)

    def policy_overlaps(policy, steps):
        return sum(1 for p in policy for spans in steps for e in spans if _overlap(p, e) > 0.005)

    policy, steps = run(SimpleEnvContainer(_envs(n, work)), 1)
    assert parallel_pairs(steps) == 0 and policy_overlaps(policy, steps) == 0  # the reference's container: all serial
    pool = AsyncEnvContainer(_envs(n, work), num_workers=4)
    try:
        policy, steps = run(pool, 1)
        assert parallel_pairs(steps) > 0                    # reward computations of one step run side by side
        assert policy_overlaps(policy, steps) == 0          # ... but the policy still waits for all of them
        policy, steps = run(pool, 2)
        assert parallel_pairs(steps) > 0
        assert policy_overlaps(policy, steps) > 0           # group B's policy evaluation hides behind group A's step
    finally:
        pool.close()
