"""Observations of many environment steps held as arrays.

The reference passes observations around as Python tuples -- ``(canvas, bag)`` with ``canvas`` a tuple of
``(label_index, (x, y, z))`` items and ``bag`` a tuple of counts (spaces.py:10-16,85-93) -- and a rollout buffer is a
list of them (buffer.py:31).  That is the interface `step()` keeps.  Between processes, however, a rollout of
256 environments x 40 canvas slots per step is tens of thousands of small Python objects per iteration: pickling it
through ``all_gather_object`` costs more than the policy update it feeds.  `ParsedObservations` is the same
information as three arrays; it behaves like the list (``len``, integer indexing gives the reference's tuple back,
index arrays give a sub-container), the parsers of the HIP agents take it without touching Python objects, and it
travels as ONE float64 matrix (`to_matrix` / `from_matrix`; labels and counts are small integers, exact in float64).
"""
from collections.abc import Sequence
from typing import List

import numpy as np


_NATIVE = False


def _native_parser():
    """the C traversal (built by __graft_entry__.build() next to the package: molgym_amd/_obsparse<EXT_SUFFIX>), or None"""
    global _NATIVE
    if _NATIVE is False:
        try:
            from molgym_amd import _obsparse
            _NATIVE = _obsparse
        except Exception:  # not built, or built for another interpreter: the numpy path serves
            _NATIVE = None
    return _NATIVE


class ParsedObservations(Sequence):
    def __init__(self, labels: np.ndarray, xyz: np.ndarray, bags: np.ndarray):
        T = labels.shape[0]
        assert labels.ndim == 2 and xyz.shape == (T, labels.shape[1], 3) and bags.ndim == 2 and bags.shape[0] == T
        self.labels = np.ascontiguousarray(labels, dtype=np.int64)   # (T, N) label index per canvas slot (0 = null)
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float64)       # (T, N, 3) as the environment holds them
        self.bags = np.ascontiguousarray(bags, dtype=np.int64)       # (T, Z)

    @classmethod
    def from_list(cls, observations: List, canvas_size: int = None, num_labels: int = None) -> 'ParsedObservations':
        if isinstance(observations, ParsedObservations):
            return observations
        T = len(observations)
        if T == 0:
            assert canvas_size is not None and num_labels is not None
            return cls(np.zeros((0, canvas_size), np.int64), np.zeros((0, canvas_size, 3)), np.zeros((0, num_labels), np.int64))
        native = _native_parser()
        if native is not None:
            # one C traversal of the tuples into preallocated arrays (csrc/obsparse.c: ~0.3 us per sample against ~1.7 for the
            # three list comprehensions + np.array below); any irregularity falls through to the numpy path and ITS messages
            try:
                first = observations[0]
                n, z = len(first[0]), len(first[1])
                labels, xyz = np.empty((T, n), np.int64), np.empty((T, n, 3), np.float64)
                bags = np.empty((T, z), np.int64)
                native.parse(observations, labels, xyz, bags)
                return cls(labels, xyz, bags)
            except (ValueError, TypeError, IndexError, OverflowError):
                pass
        try:
            labels = np.array([[item[0] for item in obs[0]] for obs in observations], dtype=np.int64)
            xyz = np.array([[item[1] for item in obs[0]] for obs in observations], dtype=np.float64)
            bags = np.array([obs[1] for obs in observations], dtype=np.int64)
        except ValueError as exc:  # ragged input
            raise RuntimeError(f'malformed observations: {exc}')
        if labels.ndim != 2 or xyz.shape != labels.shape + (3, ) or bags.ndim != 2:
            raise RuntimeError(f'malformed observations: canvas {labels.shape}, positions {xyz.shape}, bags {bags.shape}')
        return cls(labels, xyz, bags)

    # -- list behaviour ---------------------------------------------------------------------------------------------
    def __len__(self) -> int:
        return self.labels.shape[0]

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            canvas = tuple((int(l), (float(p[0]), float(p[1]), float(p[2]))) for l, p in zip(self.labels[i], self.xyz[i]))
            return canvas, tuple(int(b) for b in self.bags[i])
        return self.take(np.arange(len(self))[i] if isinstance(i, slice) else i)

    def take(self, indices) -> 'ParsedObservations':
        idx = np.asarray(indices, dtype=np.int64)
        return ParsedObservations(self.labels[idx], self.xyz[idx], self.bags[idx])

    # -- one matrix for the wire --------------------------------------------------------------------------------------
    @property
    def canvas_size(self) -> int:
        return self.labels.shape[1]

    @property
    def num_labels(self) -> int:
        return self.bags.shape[1]

    def to_matrix(self) -> np.ndarray:
        """(T, 4 N + Z) float64: [labels | xyz | bag]"""
        T, N = self.labels.shape
        return np.concatenate([self.labels.astype(np.float64), self.xyz.reshape(T, 3 * N), self.bags.astype(np.float64)], axis=1)

    @classmethod
    def from_matrix(cls, mat: np.ndarray, canvas_size: int, num_labels: int) -> 'ParsedObservations':
        T, N = mat.shape[0], canvas_size
        assert mat.shape[1] == 4 * N + num_labels
        return cls(np.rint(mat[:, :N]).astype(np.int64), mat[:, N:4 * N].reshape(T, N, 3),
                   np.rint(mat[:, 4 * N:]).astype(np.int64))

    @staticmethod
    def concatenate(parts: List['ParsedObservations']) -> 'ParsedObservations':
        return ParsedObservations(np.concatenate([p.labels for p in parts]), np.concatenate([p.xyz for p in parts]),
                                  np.concatenate([p.bags for p in parts]))
