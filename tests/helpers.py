"""Shared test helpers: build the HIP agent + the oracle with identical weights, and express oracle
intermediates in the compact (ragged) layout the kernels use."""
import ctypes as C

import numpy as np
import torch

from molgym_amd.spaces import ActionSpace, ObservationSpace
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS
from oracle.covariant_ref import CovariantACRef


def make_pair(cfg_name='cfg2', seed=0, beta='cfg', device='cuda:0', dtype=torch.float64, **overrides):
    from molgym_amd.agents.covariant import CovariantAC
    cfg = dict(CONFIGS[cfg_name])
    if beta != 'cfg':
        cfg['beta'] = beta
    kw = dict(MODEL_DEFAULTS)
    kw.update(overrides)
    torch.manual_seed(seed)
    ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']),
                     bag_scale=cfg['bag_scale'], beta=cfg['beta'], device=device, **kw)
    # non-trivial biases / widths so that every gradient path is exercised
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed + 1)
        for name, (off, shape) in ac.slot_table.items():
            n = int(np.prod(shape))
            if name.startswith('phi_') and name.endswith('bias'):
                ac.theta[off:off + n] = (0.1 * torch.randn(n, generator=g)).to(ac.theta)
            if name == 'distance_log_stds':
                ac.theta[off:off + n] += (0.2 * torch.randn(n, generator=g)).to(ac.theta)
    ref = CovariantACRef(zs=cfg['zs'], canvas_size=cfg['canvas_size'], bag_scale=cfg['bag_scale'], beta=cfg['beta'],
                         **kw).to(dtype)
    ref.load_state_dict({k: v.to(dtype).cpu() for k, v in ac.export_state_dict().items()})
    return ac, ref, cfg


def rel_err(got, want, floor=1.0):
    got = torch.as_tensor(got).double().cpu()
    want = torch.as_tensor(want).double().cpu()
    return ((got - want).abs() / want.abs().clamp(min=floor)).max().item()


def compact_vec(parts, atom_mask):
    """oracle SO3Vec (B, N, C, 2l+1, 2) -> list over l of [TA*(2l+1), 2C]."""
    out = []
    for p in parts:
        sel = p[atom_mask]  # (TA, C, m, 2)
        out.append(sel.permute(0, 2, 1, 3).reshape(sel.shape[0] * sel.shape[2], -1))
    return out


def compact_edges(parts, edge_mask):
    """oracle SO3Scalar (B, N, N, C, 2) -> list over l of [TE, 2C]."""
    return [p[edge_mask].reshape(int(edge_mask.sum()), -1) for p in parts]
