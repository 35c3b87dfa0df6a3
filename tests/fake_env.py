"""Test doubles for the rollout side: an environment with the MolecularEnvironment step/reset contract
(/root/reference/molgym/environment.py:48-79,131-142) whose reward burns a controllable amount of host CPU (the PM6
single points of reward.py:36-55 are three such calls), and a CPU policy with the agent's step(obs) contract."""
import hashlib
import time

import numpy as np
import torch


def burn(seconds: float) -> float:
    """spin (holding the GIL, like a C extension that does not release it) for `seconds`"""
    t0, x = time.perf_counter(), 0.0
    while time.perf_counter() - t0 < seconds:
        x += 1e-9
    return x


class FakeMolEnv:
    def __init__(self, canvas_size, zs, formula_counts, work_seconds=0.0, min_reward=-0.6):
        self.N, self.zs, self.formula, self.work, self.min_reward = canvas_size, list(zs), tuple(formula_counts), \
            work_seconds, min_reward
        self.reset()

    def _obs(self):
        canvas = list(self.atoms) + [(0, (0.0, 0.0, 0.0))] * (self.N - len(self.atoms))
        return tuple(canvas), tuple(self.bag)

    def reset(self):
        self.atoms, self.bag = [], list(self.formula)
        return self._obs()

    def step(self, action):
        element, position = action
        if self.zs[element] == 0:
            return self._obs(), 0.0, True, {}
        if self.bag[element] <= 0 or any(np.linalg.norm(np.subtract(position, p)) < 0.6 for _, p in self.atoms):
            return self._obs(), self.min_reward, True, {}
        t0 = time.perf_counter()  # CLOCK_MONOTONIC: comparable across the worker processes
        burn(self.work)
        t1 = time.perf_counter()
        digest = hashlib.sha256(repr((self.atoms, element, tuple(np.round(position, 6)))).encode()).digest()
        reward = (int.from_bytes(digest[:4], 'little') / 2**32 - 0.3)  # deterministic pseudo energy gain
        self.atoms.append((element, tuple(float(x) for x in position)))
        self.bag[element] -= 1
        done = len(self.atoms) == self.N or sum(self.bag) == 0
        return self._obs(), reward, done, {'elapsed_time': self.work, 'span': (t0, t1)}


class FakeAC(torch.nn.Module):
    """step(obs) -> deterministic pseudo-random valid actions (a function of the observation only)."""

    def __init__(self, zs, gpu_seconds=0.0):
        super().__init__()
        self.zs, self.gpu_seconds = list(zs), gpu_seconds
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.training = True
        self.spans = []  # (start, end) of every policy evaluation

    def step(self, observations, actions=None):
        t0 = time.perf_counter()
        time.sleep(self.gpu_seconds)  # a device-side policy evaluation: the host thread is idle meanwhile
        self.spans.append((t0, time.perf_counter()))
        rows, acts = [], []
        for canvas, bag in observations:
            rng = np.random.default_rng(int.from_bytes(hashlib.sha256(repr((canvas, bag)).encode()).digest()[:8], 'little'))
            atoms = [xyz for label, xyz in canvas if self.zs[label] != 0]
            focus = int(rng.integers(0, max(len(atoms), 1)))
            element = int(rng.choice([i for i, c in enumerate(bag) if c > 0]))
            d, o = rng.uniform(1.0, 1.8), rng.normal(size=3)
            o /= np.linalg.norm(o)
            pos = tuple(np.asarray(atoms[focus]) + d * o) if atoms else (0.0, 0.0, 0.0)
            rows.append([focus, element, d, *o])
            acts.append((element, pos))
        B = len(observations)
        a = torch.tensor(rows, dtype=torch.float32)
        return {'actions': acts, 'a': a, 'logp': -a[:, 2], 'ent': torch.zeros(B), 'v': a[:, 3] * 0.1}
