"""Persistent device canvases of the rollout: `step_canvas` draws exactly what `step(observations)`
draws, and after k environment steps the canvases in HBM equal, bit for bit, what parsing the environments' own
observations gives (float64 positions, their float32 mirror, atomic numbers, bags, atom counts)."""
import numpy as np
import pytest
import torch

from molgym_amd import ppo
from molgym_amd.buffer import PPOBufferContainer
from molgym_amd.env_container import SimpleEnvContainer
from tests.fake_env import FakeMolEnv
from tests.helpers import make_pair

pytestmark = pytest.mark.gpu
ZS = [0, 9, 16]


def _envs(n, N=7):
    return SimpleEnvContainer([FakeMolEnv(N, ZS, (0, 1 + i % 3, 2 + i % 2)) for i in range(n)])


def test_step_canvas_equals_step_and_canvas_tracks_the_environments(built_lib):
    ac, ref, cfg = make_pair('cfg2', seed=41)
    envs = _envs(12)
    obs = envs.reset()
    canvas = ac.make_canvas(obs)
    assert canvas.matches(obs)
    for it in range(9):
        torch.manual_seed(100 + it)
        with torch.no_grad():
            want = ac.step(obs)
        torch.manual_seed(100 + it)
        with torch.no_grad():
            got = ac.step_canvas(canvas)
        assert torch.equal(want['a'], got['a'])
        for k in ('logp', 'ent', 'v'):
            assert torch.equal(want[k], got[k]), k
        for (e1, p1), (e2, p2) in zip(want['actions'], got['actions']):
            assert e1 == e2 and tuple(float(x) for x in p1) == p2  # float64 positions, bit for bit
        next_obs, rewards, terminals, _ = envs.step(got['actions'])
        # environments that went on: the device already holds their new canvas
        alive = np.nonzero(~np.asarray(terminals))[0]
        if len(alive):
            assert canvas.matches([next_obs[i] for i in alive], alive)
        obs = envs.reset_if_terminal(next_obs, terminals)
        stale = canvas.stale_rows(obs, terminals)
        assert set(np.nonzero(terminals)[0]) <= set(stale)
        canvas.sync(stale, [obs[i] for i in stale])
        assert canvas.matches(obs)
    # a value-only call does not touch the canvases
    with torch.no_grad():
        ac.step_canvas(canvas, commit=False)
    assert canvas.matches(obs)


def test_rollout_with_canvas_equals_rollout_with_parsing(built_lib):
    """batch_rollout takes the canvas path for CovariantAC on the GPU; same seeds -> the same buffers as the parse path"""
    ac, ref, cfg = make_pair('cfg2', seed=42)
    bufs = []
    orig = ppo._use_canvas
    for use in (True, False):
        envs = _envs(6)
        cont = PPOBufferContainer(size=6, gamma=0.99, lam=0.97)
        torch.manual_seed(7)
        ppo._use_canvas = orig if use else (lambda agent: False)
        try:
            ppo.batch_rollout(ac, envs, cont, num_steps=6 * 8)
        finally:
            ppo._use_canvas = orig
        bufs.append(cont.merge())
    a, b = bufs
    assert a.obs_buf == b.obs_buf and a.term_buf == b.term_buf
    for f in ('act_buf', 'rew_buf', 'val_buf', 'logp_buf', 'adv_buf', 'ret_buf'):
        assert np.array_equal(np.asarray(getattr(a, f), dtype=np.float64), np.asarray(getattr(b, f), dtype=np.float64)), f


def test_pipelined_async_rollout_with_group_canvases_equals_serial_and_is_faster(built_lib):
    """SURVEY 8(f) rank 3 end to end on the GPU (BASELINE configs[4]: reward on the host CPU overlapped with the policy):
    `AsyncEnvContainer` (worker processes, step_async / step_wait of env_container.py:11-74) + CovariantAC on the HIP path +
    a reward that burns host CPU like the three PM6 single points of reward.py:36-55.  The pipelined rollout keeps one set
    of device canvases per group, steps group A's environments on the host while the GPU samples for group B, and --
    because the groups share each step's seed and key their random streams by environment id -- fills EXACTLY the buffers
    of the serial canvas rollout over the in-process container, in less wall time."""
    import time

    from molgym_amd.env_container import AsyncEnvContainer
    ac, ref, cfg = make_pair('cfg2', seed=43)
    work = 0.004  # seconds of host CPU per environment step
    mk = lambda: [FakeMolEnv(7, ZS, (0, 1 + i % 3, 2 + i % 2), work_seconds=work) for i in range(16)]
    results = {}
    for kind in ('serial', 'pipelined'):
        if kind == 'serial':
            envs = SimpleEnvContainer(mk())
        else:  # HIP is initialised in this process: the workers must not be forked from it
            envs = AsyncEnvContainer(mk(), num_workers=4, start_method='forkserver')
            assert [g[0] for g in envs.groups(2)] == [0, 1]
        cont = PPOBufferContainer(size=16, gamma=0.99, lam=0.97)
        ppo.batch_rollout(ac, envs, cont, num_steps=16 * 2)  # warm-up (workspaces, workers)
        cont = PPOBufferContainer(size=16, gamma=0.99, lam=0.97)
        envs.reset()
        torch.manual_seed(11)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ppo.batch_rollout(ac, envs, cont, num_steps=16 * 10, pipeline=2)
        torch.cuda.synchronize()
        results[kind] = (cont.merge(), time.perf_counter() - t0)
        if kind == 'pipelined':
            envs.close()
    (a, ta), (b, tb) = results['serial'], results['pipelined']
    assert a.obs_buf == b.obs_buf and a.term_buf == b.term_buf
    for f in ('act_buf', 'rew_buf', 'val_buf', 'logp_buf', 'adv_buf', 'ret_buf'):
        assert np.array_equal(np.asarray(getattr(a, f), dtype=np.float64), np.asarray(getattr(b, f), dtype=np.float64)), f
    # serial: 16 environments x 4 ms on one core per step; pipelined: 4 worker processes, and the policy evaluation of one
    # group under the other group's environment step
    assert tb < 0.75 * ta, (ta, tb)
