"""Observation / action spaces (gym- and ase-free mirror of /root/reference/molgym/spaces.py).

Only what the PPO hot path reads is kept: ``zs`` (atomic numbers, ``zs[0] == 0`` is
the null symbol), the canvas size, and the observation tuple layout
``(canvas, bag)`` with ``canvas = ((label, (x, y, z)),) * canvas_size`` and
``bag = (count,) * len(zs)`` (spaces.py:10-16,55-74,85-93).
"""
from typing import List, Sequence, Tuple

CanvasItemType = Tuple[int, Tuple[float, float, float]]
ActionType = CanvasItemType
ObservationType = Tuple[Tuple[CanvasItemType, ...], Tuple[int, ...]]


class CanvasItemSpace:
    def __init__(self, zs: List[int]) -> None:
        self.zs = list(zs)


ActionSpace = CanvasItemSpace


class CanvasSpace:
    def __init__(self, size: int, zs: List[int]) -> None:
        assert 0 in zs, '0 has to be in the list of atomic numbers'
        self.size = size
        self.zs = list(zs)


class BagSpace:
    def __init__(self, zs: List[int]) -> None:
        self.zs = list(zs)
        self.size = len(zs)


class ObservationSpace:
    def __init__(self, canvas_size: int, zs: List[int]) -> None:
        self.zs = list(zs)
        self.canvas_space = CanvasSpace(size=canvas_size, zs=zs)
        self.bag_space = BagSpace(zs=zs)

    def parse_positions(self, observation: ObservationType):
        """(positions of the non-null canvas items, in order; bag as (z, count) pairs)"""
        canvas, bag = observation
        atoms = [(self.zs[label], xyz) for label, xyz in canvas if self.zs[label] != 0]
        return atoms, tuple(zip(self.zs, bag))
