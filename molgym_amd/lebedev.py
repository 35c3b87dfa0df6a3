"""Constant table for the SO(3) normaliser: Y_lm of the Lebedev-71 points.

The reference re-evaluates the harmonics of the 1730 quadrature points on every
call (/root/reference/molgym/agents/covariant/spherical_dists.py:208-215); they
are constants, so they are computed once on the host in float64 and kept in HBM.
quadpy 0.16.2's ``lebedev_071`` is the degree-71 Lebedev rule with weights
summing to 1; scipy's ``lebedev_rule(71)`` is the same rule with weights summing
to 4 pi (point order may differ, which a logsumexp does not see).
"""
import math

import numpy as np

NLEB = 1730


def _ylm_qm(xyz: np.ndarray) -> np.ndarray:
    """Standard complex Y_l^m, l <= 4, of unit vectors (n, 3) -> (n, 25) complex128."""
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    cp = x + 1j * y
    q = {(0, 0): 1 + 0 * z, (1, 0): z, (1, 1): -1 + 0 * z, (2, 0): (3 * z * z - 1) / 2, (2, 1): -3 * z,
         (2, 2): 3 + 0 * z, (3, 0): (5 * z**3 - 3 * z) / 2, (3, 1): -(15 * z * z - 3) / 2, (3, 2): 15 * z,
         (3, 3): -15 + 0 * z, (4, 0): (35 * z**4 - 30 * z * z + 3) / 8, (4, 1): -2.5 * (7 * z**3 - 3 * z),
         (4, 2): 7.5 * (7 * z * z - 1), (4, 3): -105 * z, (4, 4): 105 + 0 * z}
    out = np.zeros((xyz.shape[0], 25), dtype=np.complex128)
    for l in range(5):
        for m in range(0, l + 1):
            nlm = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m))
            val = nlm * q[(l, m)] * cp**m
            out[:, l * l + l + m] = val
            if m:
                out[:, l * l + l - m] = (-1)**m * np.conj(val)
    return out


def _antipodal_order(pts: np.ndarray, w: np.ndarray):
    """Reorder the rule so that point g + NLEB/2 is exactly -point g (the rule is inversion symmetric).
    Y_lm(-x) = (-1)^l Y_lm(x): the fused heads kernels read only the first half of the table and split the
    coefficient sums by the parity of l; every other consumer sums over all points, where order is irrelevant."""
    n = pts.shape[0]
    key = {tuple(np.round(p, 12)): i for i, p in enumerate(pts)}
    first, second, seen = [], [], set()
    for i, p in enumerate(pts):
        if i in seen:
            continue
        j = key[tuple(np.round(-p, 12) + 0.0)]
        assert j != i and j not in seen and abs(w[i] - w[j]) <= 1e-15 * abs(w[i])
        seen.update((i, j))
        first.append(i)
        second.append(j)
    assert len(first) == n // 2
    out = np.concatenate([pts[first], -pts[first]])  # exact negation, not the tabulated partner
    return out, np.concatenate([w[first], w[first]])


def lebedev_table() -> np.ndarray:
    """float32 [51][1730]: rows 2q / 2q+1 = Re / Im Y_q, row 50 = log weight."""
    from scipy.integrate import lebedev_rule
    pts, w = lebedev_rule(71)
    assert pts.shape == (3, NLEB)
    pts, w = _antipodal_order(np.ascontiguousarray(pts.T), w)
    y = _ylm_qm(pts)
    tab = np.empty((51, NLEB), dtype=np.float64)
    tab[0:50:2] = y.real.T
    tab[1:50:2] = y.imag.T
    tab[50] = np.log(w / (4 * math.pi))
    return tab.astype(np.float32)
