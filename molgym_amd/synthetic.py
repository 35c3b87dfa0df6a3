"""Synthetic PPO mini-batches of the shape MolGym's rollout produces.

Observation tuples follow /root/reference/molgym/spaces.py:10-16,55-74,85-93:
``(canvas, bag)`` with ``canvas`` = ``canvas_size`` items ``(label, (x, y, z))``
(label indexes ``zs``; 0 is the null symbol, stored at the origin) and ``bag`` =
``len(zs)`` counts.  The recipe is SURVEY.md section 8(d): random-walk placement
with bond lengths U(1.10, 2.10) A, pairs closer than 0.6 A rejected, atom count
U{0..N} with n = 0 and n = N both present.
"""
from typing import Dict, List, Sequence

import numpy as np

CONFIGS = {
    # BASELINE.json configs[1]: SF6 single bag, covariant, canvas 7, mini-batch 140
    'cfg2': dict(zs=[0, 9, 16], canvas_size=7, batch=140, bag_scale=5, beta=-10.0),
    # configs[2]: C3H5NO3 multi-bag, canvas 12, mini-batch 1024
    'cfg3': dict(zs=[0, 1, 6, 7, 8], canvas_size=12, batch=1024, bag_scale=5, beta=-10.0),
    # configs[3]: stochastic bags, canvas 20
    'cfg4': dict(zs=[0, 1, 6, 7, 8], canvas_size=20, batch=1024, bag_scale=10, beta=-10.0),
    # configs[4]: solvation, canvas 40
    'cfg5': dict(zs=[0, 1, 6, 7, 8], canvas_size=40, batch=2048, bag_scale=20, beta=-10.0),
}

MODEL_DEFAULTS = dict(min_max_distance=(0.8, 1.8), network_width=128, maxl=4, num_cg_levels=3,
                      num_channels_hidden=10, num_channels_per_element=4, num_gaussians=3)


def _unit(rng, n=None):
    v = rng.normal(size=(3, ) if n is None else (n, 3))
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def make_canvas(rng, n_atoms: int, canvas_size: int, num_zs: int):
    pos: List[np.ndarray] = []
    while len(pos) < n_atoms:
        if not pos:
            cand = np.zeros(3)
        else:
            base = pos[rng.integers(len(pos))]
            cand = base + rng.uniform(1.10, 2.10) * _unit(rng)
            if min(np.linalg.norm(cand - p) for p in pos) < 0.6:
                continue
        pos.append(cand)
    labels = rng.integers(1, num_zs, size=n_atoms)
    canvas = [(int(l), tuple(float(x) for x in p)) for l, p in zip(labels, pos)]
    canvas += [(0, (0.0, 0.0, 0.0))] * (canvas_size - n_atoms)
    return tuple(canvas)


def make_batch(batch: int, canvas_size: int, zs: Sequence[int], seed: int = 0) -> Dict[str, object]:
    """obs list, act (B, 6) f64, logp / adv / ret (B,) f64 -- the ``data`` dict
    layout of molgym/ppo.py:77-89 for one mini-batch."""
    rng = np.random.default_rng(seed)
    nz = len(zs)
    counts = rng.integers(0, canvas_size + 1, size=batch)
    if batch >= 2:
        counts[0], counts[1] = 0, canvas_size
    obs, act = [], np.zeros((batch, 6), dtype=np.float64)
    for b in range(batch):
        n = int(counts[b])
        canvas = make_canvas(rng, n, canvas_size, nz)
        bag = rng.integers(0, 4, size=nz)
        bag[0] = 0
        if bag[1:].sum() == 0:
            bag[rng.integers(1, nz)] = 1
        obs.append((canvas, tuple(int(x) for x in bag)))
        valid = np.nonzero(bag > 0)[0]
        act[b, 0] = rng.integers(0, max(n, 1))
        act[b, 1] = rng.choice(valid)
        act[b, 2] = rng.uniform(1.1, 2.1)
        act[b, 3:6] = _unit(rng)
    adv = rng.normal(size=batch)
    adv = (adv - adv.mean()) / adv.std()
    return dict(obs=obs, act=act, logp=rng.normal(-5.0, 1.0, size=batch), adv=adv,
                ret=rng.normal(0.0, 0.3, size=batch))


def make_batch_internal(batch: int, canvas_size: int, zs: Sequence[int], seed: int = 0) -> Dict[str, object]:
    """Same canvases, with the 7-column internal-coordinate actions of SchNetAC
    (stop, focus, element, distance, angle, dihedral, kappa; internal/agent.py:26,306-308)."""
    d = make_batch(batch, canvas_size, zs, seed)
    rng = np.random.default_rng(seed + 1000)
    a6 = d['act']
    act = np.zeros((batch, 7), dtype=np.float64)
    act[:, 1], act[:, 2], act[:, 3] = a6[:, 0], a6[:, 1], a6[:, 2]
    act[:, 4] = rng.uniform(0.3, np.pi - 0.3, size=batch)
    act[:, 5] = rng.uniform(0.2, np.pi - 0.2, size=batch)
    act[:, 6] = rng.integers(0, 2, size=batch)
    d['act'] = act
    return d
