"""Vectorised environments (mirror of /root/reference/molgym/env_container.py:11-128) plus the process-pool container
BASELINE configs[4] asks for.

`VecEnv` / `SimpleEnvContainer` keep the reference's interface (``reset``, ``step_async``, ``step_wait``, ``step``,
``get_size``, ``reset_if_terminal``).  The reference declares ``step_async`` / ``step_wait`` but its only container
steps the environments one after another inside ``step_wait`` (env_container.py:84-97): every environment step is
three PM6 single points on the host (reward.py:36-55), so a 256-environment rollout step is 768 serial Sparrow calls
during which the GPU idles.

`AsyncEnvContainer` really is asynchronous: the environments live in worker PROCESSES (the reward code holds the GIL
and Sparrow state), ``step_async`` only posts the actions and returns, ``step_wait`` collects.  Two things follow:
the environments of one step run in parallel on the host cores, and -- because ``step_async`` may be given a SUBSET
of the environments -- `ppo.batch_rollout` can keep one group of environments stepping on the CPU while the GPU
evaluates the policy for the other group (double buffering).  Environments are independent (env_container.py:91-97),
so the trajectories are the same as the serial container's.
"""
import multiprocessing as mp
import os
import traceback
from abc import ABC, abstractmethod
from typing import List, Optional, Sequence, Tuple

import numpy as np


class VecEnv(ABC):
    @abstractmethod
    def reset(self) -> List:
        raise NotImplementedError

    @abstractmethod
    def step_async(self, actions) -> None:
        raise NotImplementedError

    @abstractmethod
    def step_wait(self) -> Tuple[List, np.ndarray, np.ndarray, List[dict]]:
        raise NotImplementedError

    def step(self, actions) -> Tuple[List, np.ndarray, np.ndarray, List[dict]]:
        self.step_async(actions)
        return self.step_wait()

    def render(self, mode='human'):
        raise NotImplementedError

    @abstractmethod
    def get_size(self) -> int:
        raise NotImplementedError

    @abstractmethod
    def reset_if_terminal(self, observations: List, terminals: List[bool]):
        raise NotImplementedError

    def close(self):
        pass


class SimpleEnvContainer(VecEnv):
    """In-process, serial (the reference's container)."""

    def __init__(self, environments: List):
        self.environments = environments
        self.actions = None

    def step_async(self, actions) -> None:
        self.actions = actions

    def step_wait(self):
        assert self.actions is not None and len(self.environments) == len(self.actions)
        results = [env.step(action) for env, action in zip(self.environments, self.actions)]
        self.actions = None
        obs, rewards, dones, infos = zip(*results)
        return list(obs), np.array(rewards), np.array(dones), list(infos)

    def reset(self):
        return [env.reset() for env in self.environments]

    def reset_if_terminal(self, observations, terminals):
        assert len(self.environments) == len(observations) == len(terminals)
        return [env.reset() if done else obs for env, obs, done in zip(self.environments, observations, terminals)]

    def get_size(self) -> int:
        return len(self.environments)


def _worker(conn, environments):
    """Owns `environments`; serves ('step', [(slot, action)]) / ('reset', [slot]) requests until 'close'."""
    try:
        while True:
            cmd, payload = conn.recv()
            if cmd == 'step':
                conn.send(('ok', [environments[slot].step(action) for slot, action in payload]))
            elif cmd == 'reset':
                conn.send(('ok', [environments[slot].reset() for slot in payload]))
            elif cmd == 'close':
                conn.send(('ok', None))
                return
            else:
                conn.send(('error', f'unknown command {cmd!r}'))
    except (EOFError, KeyboardInterrupt):
        return
    except Exception:  # surface the worker's traceback in the parent instead of dying silently
        try:
            conn.send(('error', traceback.format_exc()))
        except Exception:
            pass


class AsyncEnvContainer(VecEnv):
    """Environments distributed round-robin over `num_workers` processes.

    ``step_async(actions, indices=None)`` posts the actions of the environments `indices` (default: all) and returns
    a ticket at once; ``step_wait(ticket=None)`` blocks until that request (default: the oldest pending one) is
    complete and returns ``(observations, rewards, dones, infos)`` in the order of `indices`.  Several requests may be
    pending at the same time as long as they involve disjoint WORKERS (a worker answers its pipe in order);
    ``groups(k)`` partitions the environments accordingly.

    start_method 'fork' (default) ships the already constructed environments to the workers by inheritance and
    needs no pickling; create the container BEFORE the first HIP call of the process or use 'spawn' /
    'forkserver' (the environments must then be picklable) -- the workers never touch the GPU either way.
    """

    def __init__(self, environments: List, num_workers: Optional[int] = None, start_method: str = 'fork'):
        n = len(environments)
        assert n > 0
        self.size = n
        self.num_workers = max(1, min(n, num_workers or os.cpu_count() or 1))
        ctx = mp.get_context(start_method)
        self._home = [(i % self.num_workers, i // self.num_workers) for i in range(n)]  # env -> (worker, slot)
        self._conns, self._procs = [], []
        for w in range(self.num_workers):
            parent, child = ctx.Pipe()
            proc = ctx.Process(target=_worker, args=(child, environments[w::self.num_workers]), daemon=True)
            proc.start()
            child.close()
            self._conns.append(parent)
            self._procs.append(proc)
        self._pending = []   # tickets in issue order: (ticket id, indices, [(worker, positions)])
        self._next_ticket = 0
        self._busy = set()   # workers with a step in flight
        self._closed = False

    def groups(self, k: int) -> List[List[int]]:
        """At most k groups of environments living on disjoint sets of workers (worker w -> group w mod k): requests on
        different groups never share a pipe, so one group can be stepping while another is reset or collected."""
        k = max(1, min(k, self.num_workers))
        out = [[i for i in range(self.size) if self._home[i][0] % k == g] for g in range(k)]
        return [g for g in out if g]

    # -- plumbing ---------------------------------------------------------------------------------------------------
    def _scatter(self, cmd: str, indices: Sequence[int], payloads: Optional[Sequence] = None):
        per_worker = {}
        for pos, i in enumerate(indices):
            w, slot = self._home[i]
            per_worker.setdefault(w, ([], []))
            per_worker[w][0].append(pos)
            per_worker[w][1].append(slot if payloads is None else (slot, payloads[pos]))
        if self._busy & set(per_worker):
            raise RuntimeError('request on a worker whose previous step has not been collected (see groups())')
        for w, (_, items) in per_worker.items():
            self._conns[w].send((cmd, items))
        return [(w, positions) for w, (positions, _) in per_worker.items()]

    def _gather(self, plan, count: int) -> list:
        out = [None] * count
        for w, positions in plan:
            status, payload = self._conns[w].recv()
            if status != 'ok':
                raise RuntimeError(f'environment worker {w} failed:\n{payload}')
            for pos, item in zip(positions, payload):
                out[pos] = item
        return out

    # -- VecEnv -----------------------------------------------------------------------------------------------------
    def get_size(self) -> int:
        return self.size

    def reset(self, indices: Optional[Sequence[int]] = None) -> List:
        idx = list(range(self.size)) if indices is None else list(indices)
        return self._gather(self._scatter('reset', idx), len(idx))

    def step_async(self, actions, indices: Optional[Sequence[int]] = None) -> int:
        idx = list(range(self.size)) if indices is None else list(indices)
        assert len(actions) == len(idx)
        plan = self._scatter('step', idx, list(actions))
        ticket = self._next_ticket
        self._next_ticket += 1
        self._pending.append((ticket, idx, plan))
        self._busy |= {w for w, _ in plan}
        return ticket

    def step_wait(self, ticket: Optional[int] = None):
        assert self._pending, 'step_wait() without a pending step_async()'
        pos = 0 if ticket is None else [t for t, _, _ in self._pending].index(ticket)
        _, idx, plan = self._pending.pop(pos)
        results = self._gather(plan, len(idx))
        self._busy -= {w for w, _ in plan}
        obs, rewards, dones, infos = zip(*results)
        return list(obs), np.array(rewards), np.array(dones), list(infos)

    def reset_if_terminal(self, observations, terminals, indices: Optional[Sequence[int]] = None):
        idx = list(range(self.size)) if indices is None else list(indices)
        assert len(idx) == len(observations) == len(terminals)
        todo = [pos for pos, done in enumerate(terminals) if done]
        out = list(observations)
        if todo:
            fresh = self._gather(self._scatter('reset', [idx[pos] for pos in todo]), len(todo))
            for pos, obs in zip(todo, fresh):
                out[pos] = obs
        return out

    def close(self):
        if self._closed:
            return
        self._closed = True
        for conn in self._conns:
            try:
                conn.send(('close', None))
            except Exception:
                pass
        for proc in self._procs:
            proc.join(timeout=2)
            if proc.is_alive():
                proc.terminate()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
