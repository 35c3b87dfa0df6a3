"""Persistent device-side canvases for rollouts.

`CovariantAC.step(observations)` keeps the reference's contract: Python observation tuples in, parsed on every call
(covariant/agent.py:165-197).  During a rollout that parse is most of the step -- and redundant: between two steps of
an episode the canvas changes by exactly the atom the agent just placed.  A `DeviceCanvas` keeps the E environments'
canvases in HBM (float64 positions as the environment holds them, their float32 mirror for the kernels, atomic
numbers, bags, atom counts); `CovariantAC.step_canvas(canvas)` samples on the resident arrays, appends the drawn
atoms in place (`mg_canvas_append`) and returns the same dict as `step(observations)`.  The host uploads a row only
when an environment was reset (`canvas.sync`).

The environment stays the ground truth: `canvas.matches(observations)` re-parses and compares (tests, debugging)."""
import ctypes as C
from typing import List, Sequence

import numpy as np
import torch

from .. import _lib


def parse_canvases_f64(observations, zs, canvas_size):
    """(pos float64 (B, N, 3), charges int32, bags float32, natoms): like parse_observations_host but keeping the
    environment's float64 positions"""
    from .covariant import compact_canvases, observation_arrays
    labels, xyz, bags = observation_arrays(observations, zs, canvas_size)
    xyz, charges, natoms = compact_canvases(labels, xyz, zs)
    return xyz, charges, bags.astype(np.float32), natoms.astype(np.int32)


class DeviceCanvas:
    def __init__(self, ac, observations: List):
        self.ac, self.E = ac, len(observations)
        self.N, self.zs = ac.observation_space.canvas_space.size, list(ac.zs)
        dev = ac.theta.device
        pos, charges, bags, natoms = parse_canvases_f64(observations, self.zs, self.N)
        self.pos64 = torch.from_numpy(pos).to(dev)
        self.pos32 = self.pos64.float()
        self.charges = torch.from_numpy(charges).to(dev)
        self.bags = torch.from_numpy(bags).to(dev)
        self.natoms_dev = torch.from_numpy(natoms).to(dev)
        self.natoms = natoms.copy()  # host mirrors: size the kernels' ragged lists / spot foreign changes, no round trip
        self.bags_host = bags.astype(np.int64)
        self._zs_c = (C.c_int32 * len(self.zs))(*self.zs)
        self.last_placed = None  # (mask, float64 positions) of the atoms the last committed step_canvas placed

    def sync(self, indices: Sequence[int], observations: List) -> None:
        """overwrite the rows `indices` with freshly parsed observations (environments that were reset)"""
        if len(indices) == 0:
            return
        pos, charges, bags, natoms = parse_canvases_f64(observations, self.zs, self.N)
        # ONE host-to-device copy: [positions | charges | bag | atom count | row index] per environment as float64 (all
        # integers here are exact in it); the five scatters below are device-side (five separate small uploads cost 2.5 ms
        # for 20 environments, more than five rollout steps)
        k, N, Z = len(indices), self.N, len(self.zs)
        packed = np.empty((k, 4 * N + Z + 2), dtype=np.float64)
        packed[:, :3 * N] = pos.reshape(k, 3 * N)
        packed[:, 3 * N:4 * N] = charges
        packed[:, 4 * N:4 * N + Z] = bags
        packed[:, 4 * N + Z] = natoms
        packed[:, 4 * N + Z + 1] = np.asarray(indices, dtype=np.int64)
        buf = torch.from_numpy(packed).to(self.pos64.device)
        idx = buf[:, 4 * N + Z + 1].long()
        p = buf[:, :3 * N].reshape(k, N, 3)
        # index_copy_: one launch per array (advanced-indexing assignment is several launches plus index bookkeeping on the
        # host -- 2.6 ms for 20 environments, six rollout steps' worth)
        self.pos64.index_copy_(0, idx, p)
        self.pos32.index_copy_(0, idx, p.float())
        self.charges.index_copy_(0, idx, buf[:, 3 * N:4 * N].to(self.charges.dtype))
        self.bags.index_copy_(0, idx, buf[:, 4 * N:4 * N + Z].to(self.bags.dtype))
        self.natoms_dev.index_copy_(0, idx, buf[:, 4 * N + Z].to(self.natoms_dev.dtype))
        self.natoms[np.asarray(indices)] = natoms
        self.bags_host[np.asarray(indices)] = bags.astype(np.int64)
        if self.last_placed is not None:
            self.last_placed[0][np.asarray(indices)] = False

    def stale_rows(self, observations: List, terminals) -> np.ndarray:
        """environments whose canvas on the device no longer describes `observations`: the ones that were reset, any whose
        bag differs from the mirror (an environment that refills its bag, environment.py:186-196), and any whose newest
        atom is not the one the last `step_canvas` placed -- the device canvases assume APPEND-ONLY environments (the
        reference's: environment.py:95-117 appends exactly the agent's atom or terminates); an environment that relaxes or
        edits its canvas is caught here and simply re-uploaded every step.  Host-side tests on B short tuples, no parse."""
        bags = np.array([obs[1] for obs in observations], dtype=np.int64)
        stale = np.asarray(terminals, dtype=bool) | (bags != self.bags_host).any(axis=1)
        if self.last_placed is not None:
            placed, positions = self.last_placed
            for b in np.nonzero(placed & ~stale)[0]:
                item = observations[b][0][int(self.natoms[b]) - 1]
                if self.zs[item[0]] == 0 or tuple(item[1]) != tuple(positions[b]):
                    stale[b] = True
        return np.nonzero(stale)[0]

    def append(self, actions: torch.Tensor, commit: bool = True) -> torch.Tensor:
        """place the atoms of the action rows (E, 6) on the canvases (commit) or only compute where they would go;
        returns the positions, (E, 3) float64"""
        newpos = torch.empty(self.E, 3, dtype=torch.float64, device=self.pos64.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        tgt = (self.pos64, self.pos32, self.charges, self.bags, self.natoms_dev)
        if not commit:  # scratch copies take the update: a value-only call at the end of a rollout (ppo.py:205)
            tgt = tuple(t.clone() for t in tgt)
        with self.ac._guard():
            _lib.check(_lib.lib().mg_canvas_append(self.E, self.N, len(self.zs), self._zs_c, p(actions), p(tgt[0]),
                                                   p(tgt[1]), p(tgt[2]), p(tgt[3]), p(tgt[4]), p(newpos), self.ac._s()))
        return newpos

    def matches(self, observations: List, indices=None) -> bool:
        """device canvases == what parsing `observations` gives (bit for bit)"""
        pos, charges, bags, natoms = parse_canvases_f64(observations, self.zs, self.N)
        sel = slice(None) if indices is None else torch.as_tensor(np.asarray(indices), device=self.pos64.device)
        return (np.array_equal(self.pos64[sel].cpu().numpy(), pos) and
                np.array_equal(self.pos32[sel].cpu().numpy(), pos.astype(np.float32)) and
                np.array_equal(self.charges[sel].cpu().numpy(), charges) and
                np.array_equal(self.bags[sel].cpu().numpy(), bags) and
                np.array_equal(self.natoms_dev[sel].cpu().numpy(), natoms) and
                np.array_equal(self.natoms if indices is None else self.natoms[np.asarray(indices)], natoms))
