"""Trajectory buffers of the PPO rollout (mirror of /root/reference/molgym/buffer.py:10-116 and
buffer_container.py:10-81; same class names, methods, attribute names -- RolloutSaver pickles these objects and
tools/analysis.py reads the ``*_buf`` attributes back).

Two ways out of a finished buffer:

* ``get_data()`` -- the reference's: GAE-lambda per path in float64 on the host as each path finishes
  (buffer.py:54-92), population-standardised advantages without epsilon (:104-110), numpy arrays.
* ``get_data(device=...)`` -- the same numbers produced ON the device for `ppo.train`'s device path: rewards,
  values and the path table are uploaded once, ``mg_gae`` evaluates every path of the rollout in one launch and
  ``mg_adv_normalize`` standardises in place (include/molgym_hip.h), so ``adv`` / ``ret`` / ``logp`` are born in
  HBM as float64 tensors and `CovariantAC.prepare_rollout` takes them without a host round trip.
"""
import ctypes as C
import itertools
from typing import List, Optional, Tuple

import numpy as np


def discount_cumsum(x: np.ndarray, discount: float) -> np.ndarray:
    """tools/util.py:72-87: y[t] = x[t] + discount * y[t+1] (float64)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.empty_like(x)
    run = 0.0
    for t in range(len(x) - 1, -1, -1):
        run = x[t] + discount * run
        y[t] = run
    return y


class DynamicPPOBuffer:
    BUFFER_FIELDS = [
        'obs_buf', 'act_buf', 'rew_buf', 'next_obs_buf', 'term_buf', 'val_buf', 'logp_buf', 'adv_buf', 'ret_buf'
    ]

    def __init__(self, gamma=0.99, lam=0.95) -> None:
        for field in self.BUFFER_FIELDS:
            setattr(self, field, [])
        self.gamma, self.lam = gamma, lam
        self.current_index = 0
        self.start_index = 0
        # path table for the device form of get_data: (start, end, bootstrap value) of every finished path
        self.paths: List[Tuple[int, int, float]] = []

    def store(self, obs, act: np.ndarray, reward: float, next_obs, terminal: bool, value: float, logp: float) -> None:
        self.obs_buf.append(obs)
        self.act_buf.append(act)
        self.rew_buf.append(reward)
        self.next_obs_buf.append(next_obs)
        self.term_buf.append(terminal)
        self.val_buf.append(value)
        self.logp_buf.append(logp)
        self.current_index += 1

    def finish_path(self, last_val: float) -> Tuple[Optional[float], int]:
        """Close the running trajectory: delta_t = r_t + gamma V_{t+1} - V_t, adv = discount_cumsum(delta, gamma lam),
        ret = discount_cumsum(r ++ last_val, gamma)[:-1]; last_val = 0 after a terminal state, V(s_T) when the rollout
        cuts the episode.  Returns (episodic return, length) or (None, 0) when nothing is open."""
        if self.is_finished():
            return None, 0
        lo, hi = self.start_index, self.current_index
        rews = np.array(self.rew_buf[lo:hi] + [last_val], dtype=np.float64)
        vals = np.array(self.val_buf[lo:hi] + [last_val], dtype=np.float64)
        deltas = rews[:-1] + self.gamma * vals[1:] - vals[:-1]
        self.adv_buf += discount_cumsum(deltas, self.gamma * self.lam).tolist()
        self.ret_buf += discount_cumsum(rews, self.gamma).tolist()[:-1]
        self.paths.append((lo, hi, float(last_val)))
        self.start_index = hi
        assert all(len(getattr(self, f)) == hi for f in self.BUFFER_FIELDS)
        return self.ret_buf[lo], hi - lo

    def is_finished(self) -> bool:
        return self.start_index == self.current_index

    def get_data(self, device=None) -> dict:
        assert self.is_finished()
        if device is not None:
            return self._get_data_device(device)
        adv = np.array(self.adv_buf)
        return dict(obs=self.obs_buf, act=np.array(self.act_buf), ret=np.array(self.ret_buf),
                    adv=(adv - np.mean(adv)) / np.std(adv), logp=np.array(self.logp_buf))

    def _get_data_device(self, device) -> dict:
        import torch
        from . import _lib
        lib = _lib.lib()
        dev = torch.device(device)
        T, P = self.current_index, len(self.paths)
        off = np.array([p[0] for p in self.paths] + [T], dtype=np.int32)
        assert P > 0 and off[0] == 0 and all(self.paths[i][1] == off[i + 1] for i in range(P))
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        rew, val = up(self.rew_buf, np.float64), up(self.val_buf, np.float64)
        last, d_off = up([p[2] for p in self.paths], np.float64), up(off, np.int32)
        adv = torch.empty(T, dtype=torch.float64, device=dev)
        ret = torch.empty(T, dtype=torch.float64, device=dev)
        scratch = torch.empty(2, dtype=torch.float64, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.mg_gae(P, p(d_off), p(rew), p(val), p(last), float(self.gamma), float(self.lam), p(adv),
                                  p(ret), s))
            _lib.check(lib.mg_adv_normalize(T, p(adv), p(scratch), s))
        return dict(obs=self.obs_buf, act=np.array(self.act_buf), ret=ret, adv=adv, logp=up(self.logp_buf, np.float64))


class PPOBufferContainer:
    """One buffer per environment of the VecEnv (buffer_container.py:10-81)."""

    def __init__(self, size: int, gamma: float, lam: float) -> None:
        self.gamma, self.lam, self.size = gamma, lam, size
        self.buffers = [DynamicPPOBuffer(gamma=gamma, lam=lam) for _ in range(size)]
        self.episodic_returns: List[float] = []
        self.episode_lengths: List[int] = []

    def get_num_episodes(self) -> int:
        assert len(self.episodic_returns) == len(self.episode_lengths)
        return len(self.episodic_returns)

    def store(self, observations, actions: np.ndarray, rewards: np.ndarray, next_observations,
              terminals: np.ndarray, values: np.ndarray, logps: np.ndarray, env_indices=None) -> None:
        """`env_indices` (extension): the buffers these rows belong to -- the pipelined rollout steps the
        environments in groups; None = all of them, in order, like the reference."""
        idx = range(self.size) if env_indices is None else env_indices
        n = len(idx)
        assert (len(observations) == actions.shape[0] == rewards.shape[0] == len(next_observations) ==
                terminals.shape[0] == values.shape[0] == logps.shape[0] == n)
        for row, i in enumerate(idx):
            buf = self.buffers[i]
            buf.store(obs=observations[row], act=actions[row], reward=rewards[row], next_obs=next_observations[row],
                      terminal=terminals[row], value=values[row], logp=logps[row])
            if terminals[row]:
                episodic_return, length = buf.finish_path(0.0)
                assert episodic_return is not None and length > 0
                self.episodic_returns.append(episodic_return)
                self.episode_lengths.append(length)

    def finish_paths(self, values: np.ndarray, env_indices=None) -> None:
        idx = range(self.size) if env_indices is None else env_indices
        assert values.shape[0] == len(idx)
        for i, value in zip(idx, values):
            if not self.buffers[i].is_finished():
                self.buffers[i].finish_path(value)  # a cut-off path is not recorded as an episode

    def merge(self) -> DynamicPPOBuffer:
        assert all(b.is_finished() for b in self.buffers)
        new = DynamicPPOBuffer(gamma=self.gamma, lam=self.lam)
        for field in DynamicPPOBuffer.BUFFER_FIELDS:
            setattr(new, field, list(itertools.chain.from_iterable(getattr(b, field) for b in self.buffers)))
        base = 0
        for b in self.buffers:
            new.paths += [(lo + base, hi + base, lv) for lo, hi, lv in b.paths]
            base += b.current_index
        new.current_index = new.start_index = base
        return new
