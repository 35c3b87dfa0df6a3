"""CovariantAC.step of the oracle (test infrastructure, see oracle/__init__.py).

Follows /root/reference/molgym/agents/covariant/agent.py:20-334 (constructor
:21-145, parse_observations :165-197, step :209-334), so3_tools.py:41-79,
108-132,141-192, spherical_dists.py:160-215,273-286, gmm.py:8-18,
tools.py:8-49 (re-done without ase), /root/reference/molgym/modules.py:8-50 and
spaces.py:55-61,106-107.  Third-party pieces restated from their published
behaviour: torch-scatter 2.0.5 scatter_softmax (eps 1e-12), quadpy 0.16.2
lebedev_071 (here scipy.integrate.lebedev_rule(71), weights / 4 pi).
"""
import math
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from . import so3
from .encoder_ref import Cormorant, CormorantMixer

_LEBEDEV = {}


def lebedev_71():
    """(points (1730, 3), weights (1730,) summing to 1) as float64 numpy."""
    if 'g' not in _LEBEDEV:
        from scipy.integrate import lebedev_rule
        x, w = lebedev_rule(71)
        _LEBEDEV['g'] = (np.ascontiguousarray(x.T), w / (4 * math.pi))
    return _LEBEDEV['g']


def to_one_hot(indices, num_classes):
    shape = tuple(indices.shape[:-1]) + (num_classes, )
    oh = torch.zeros(shape).view(-1, num_classes)
    oh.scatter_(1, indices.view(-1, 1), 1)  # RuntimeError on out-of-range index
    return oh.view(*shape)


def masked_softmax(logits, mask, eps=1e-12):
    """scatter_softmax with index = mask (two groups per row), times mask."""
    out = torch.zeros_like(logits)
    for g in (0, 1):
        sel = mask == bool(g)
        neg = torch.full_like(logits, -float('inf'))
        grp = torch.where(sel, logits, neg)
        mx = grp.max(dim=-1, keepdim=True).values
        mx = torch.where(torch.isfinite(mx), mx, torch.zeros_like(mx))
        ex = torch.where(sel, (logits - mx).exp(), torch.zeros_like(logits))
        out = out + ex / (ex.sum(dim=-1, keepdim=True) + eps)
    return out * mask


class MLP(nn.Module):
    def __init__(self, input_dim, output_dims):
        super().__init__()
        dims = (input_dim, ) + tuple(output_dims)
        self.layers = nn.ModuleList()
        for a, b in zip(dims[:-1], dims[1:]):
            lin = nn.Linear(a, b)
            nn.init.orthogonal_(lin.weight.data)
            nn.init.constant_(lin.bias.data, 0)
            self.layers.append(lin)

    def forward(self, x):
        for lin in self.layers[:-1]:
            x = torch.relu(lin(x))
        return self.layers[-1](x)


def atomic_scalars(vec, maxl):
    """l=0 part (re, im) then, per l, (sum_m (-1)^m [re re' - im im'], sum |a|^2)
    with a' = a[-m]; concatenated over channels-blocks and flattened."""
    blocks = [vec[0]]
    for l, part in zip(range(maxl + 1), vec):
        s = torch.tensor([(-1.0) ** m for m in range(-l, l + 1)], dtype=part.dtype)
        sign = torch.stack([s, -s], dim=-1)
        prod = (sign * part * part.flip(-2)).sum(dim=(-1, -2), keepdim=True)
        nrm = (part * part).sum(dim=(-1, -2), keepdim=True)
        blocks.append(torch.cat([prod, nrm], dim=-1))
    return torch.cat(blocks, dim=-3).flatten(start_dim=-3)


def normalize_alms(a_lms):
    k = sum(p.sum(dim=-3).square().sum(dim=(-1, -2)) for p in a_lms)
    root = k.clamp(min=1e-10).sqrt().view(k.shape + (1, 1, 1))
    return so3.SO3Vec([p / root for p in a_lms])


def sum_product_alms_ylms(a_lms, y_lms):
    return sum(so3.cmul(a, y).sum(dim=(-2, -3)) for a, y in zip(a_lms, y_lms))


class CategoricalRef:
    """torch.distributions.Categorical(probs=...) semantics: renormalise, logits =
    log(clamp(p, eps, 1-eps))."""

    def __init__(self, probs):
        self.probs = probs / probs.sum(-1, keepdim=True)
        eps = torch.finfo(torch.float32).eps  # the reference runs in float32
        self.logits = torch.log(self.probs.clamp(min=eps, max=1 - eps))

    def log_prob(self, value):
        return self.logits.gather(-1, value.unsqueeze(-1)).squeeze(-1)

    def entropy(self):
        return -(self.logits * self.probs).sum(-1)

    def sample(self):
        return torch.multinomial(self.probs, 1).squeeze(-1)


class GMMRef:
    def __init__(self, log_probs, means, stds):
        self.mix_logits = log_probs - log_probs.logsumexp(-1, keepdim=True)
        self.means, self.stds = means, stds

    def log_prob(self, x):
        x = x.unsqueeze(-1)
        comp = -((x - self.means)**2) / (2 * self.stds**2) - self.stds.log() - math.log(math.sqrt(2 * math.pi))
        return torch.logsumexp(comp + self.mix_logits, dim=-1)


class SO3DistRef:
    """ExpSO3Distribution when beta is given, else SO3Distribution."""

    def __init__(self, a_lms, cg, maxl, beta, empty):
        self.cg, self.maxl, self.beta, self.empty = cg, maxl, beta, empty
        self.coefficients = normalize_alms(a_lms)
        if beta is not None:
            pts, wts = lebedev_71()
            dtype = a_lms[0].dtype
            grid = torch.tensor(pts, dtype=dtype).unsqueeze(-2)  # recomputed every call, like the reference
            w = torch.tensor(wts, dtype=dtype).unsqueeze(-1)
            self.log_z = math.log(4 * math.pi) + torch.logsumexp(self._unnorm(grid) + w.log(), dim=0)

    def _s2(self, value):
        y = so3.spherical_harmonics(self.cg.to(value.dtype), value, self.maxl, True, False, 'qm')
        s = sum_product_alms_ylms(self.coefficients, y)
        return s.square().sum(-1)

    def _unnorm(self, value):
        return -self.beta * self._s2(value)

    def log_prob(self, value):
        if self.beta is not None:
            return self._unnorm(value) - self.log_z
        p = self._s2(value)
        if self.empty is not None:
            p = torch.where(self.empty, torch.full_like(p, 1 / (4 * math.pi)), p)
        return torch.log(p.clamp(min=1e-10))


def parse_observations(observations, zs, canvas_size, dtype=torch.float32):
    """Observation tuples -> tensors: null (label 0) items dropped, atoms kept in
    order and zero-padded to canvas_size; charges = zs[label]."""
    B = len(observations)
    pos = np.zeros((B, canvas_size, 3), dtype=np.float64)
    charges = np.zeros((B, canvas_size), dtype=np.int32)
    counts = np.zeros((B, ), dtype=np.int64)
    for b, (canvas, _) in enumerate(observations):
        if len(canvas) != canvas_size:
            raise RuntimeError('canvas length does not match canvas_size')
        k = 0
        for label, xyz in canvas:
            if label < 0:
                raise RuntimeError(f'Invalid atomic number: {label}')
            if zs[label] != 0:
                pos[b, k] = xyz
                charges[b, k] = zs[label]
                k += 1
        counts[b] = k
    data = {
        # the reference builds float32 tensors (covariant/tools.py:13); a float64 oracle sees the same
        # float32-quantised inputs, so it is the exact-arithmetic value of the float32 computation
        'positions': torch.tensor(pos, dtype=torch.float32).to(dtype),
        'charges': torch.tensor(charges, dtype=torch.int32),
        'num_atoms': torch.tensor(counts, dtype=torch.int32),
    }
    zs_t = torch.tensor(zs, dtype=dtype)
    data['one_hot'] = data['charges'].unsqueeze(-1) == zs_t.view(1, 1, -1)
    data['atom_mask'] = data['charges'] > 0
    data['edge_mask'] = data['atom_mask'].unsqueeze(1) * data['atom_mask'].unsqueeze(2)
    default = torch.zeros_like(data['atom_mask'])
    default[..., 0] = 1
    data['focus_mask'] = torch.logical_or(data['atom_mask'], default)
    data['empty'] = torch.tensor(counts == 0)
    data['bags'] = torch.tensor([list(o[1]) for o in observations], dtype=dtype)
    data['element_mask'] = data['bags'] > 0
    data['value_mask'] = data['atom_mask']
    return data


class CovariantACRef(nn.Module):
    def __init__(self, zs: List[int], canvas_size: int, min_max_distance, network_width, maxl, num_cg_levels,
                 num_channels_hidden, num_channels_per_element, num_gaussians, bag_scale,
                 beta: Optional[float] = None):
        super().__init__()
        self.zs, self.canvas_size = list(zs), canvas_size
        self.min_distance, self.max_distance = min_max_distance
        assert self.min_distance < self.max_distance
        self.beta, self.max_sh = beta, maxl
        self.num_channels_per_element, self.num_gaussians = num_channels_per_element, num_gaussians
        self.num_channels_out = len(zs) * num_channels_per_element
        self.cg = so3.CGTable(maxl, torch.float64)
        self.cg_model = Cormorant(maxl=maxl, max_sh=maxl, num_cg_levels=num_cg_levels,
                                  num_channels=[num_channels_hidden] * num_cg_levels + [self.num_channels_out],
                                  num_species=len(zs), soft_cut_rad=min(self.max_distance, 2.1), soft_cut_width=0.2,
                                  level_gain=10.0, charge_power=2, basis_set=[3, 3], charge_scale=max(zs),
                                  bag_scale=bag_scale, cg=self.cg)
        ce = num_channels_per_element
        self.cg_mix = CormorantMixer([ce] * (maxl + 1), [ce], maxl, ce, 10.0, self.cg)
        self.num_latent = (maxl + 2) * self.num_channels_out * 2
        self.num_latent_element = (maxl + 2) * ce * 2
        self.phi_focus = MLP(self.num_latent, (network_width, 1))
        self.phi_element = MLP(self.num_latent, (network_width, len(zs)))
        self.phi_d = MLP(self.num_latent_element, (network_width, 2 * num_gaussians))
        self.distance_log_stds = nn.Parameter(torch.log(torch.tensor([0.1] * num_gaussians)))
        self.phi_trans = MLP(self.num_latent, (network_width, network_width))
        self.phi_v = MLP(network_width, (network_width, 1))
        self.training = True

    def step(self, observations, actions=None, dtype=torch.float32, return_internals=False):
        """actions given -> evaluate (the PPO training path, ppo.py:26).  Sampling
        (actions None) is rollout-side and not restated here."""
        if actions is None:
            raise NotImplementedError('oracle covers the action-evaluation path only')
        data = parse_observations(observations, self.zs, self.canvas_size, dtype)
        actions = torch.as_tensor(actions, dtype=torch.float32).to(dtype)  # agent.py:214 casts to float32
        B, N, ce = len(observations), self.canvas_size, self.num_channels_per_element
        covariats = self.cg_model(data)
        invariats = atomic_scalars(covariats, self.max_sh)

        focus_logits = self.phi_focus(invariats).squeeze(-1)
        focus_probs = masked_softmax(focus_logits, data['focus_mask'])
        focus_dist = CategoricalRef(focus_probs)
        focus = torch.round(actions[:, :1]).long()
        focus_oh = to_one_hot(focus, N).to(dtype)
        focused_cov = so3.SO3Vec([torch.einsum('ba,batmx->btmx', focus_oh, p) for p in covariats])
        focused_inv = torch.einsum('ba,baf->bf', focus_oh, invariats)

        element_logits = self.phi_element(focused_inv)
        element_probs = masked_softmax(element_logits, data['element_mask'])
        element_dist = CategoricalRef(element_probs)
        element = torch.round(actions[:, 1:2]).long()

        idx = torch.arange(ce).unsqueeze(0) + element * ce  # (B, ce)
        element_cov = so3.SO3Vec([
            torch.gather(p, 1, idx.view(B, ce, 1, 1).expand(-1, -1, p.shape[-2], 2)) for p in focused_cov
        ])
        element_inv = atomic_scalars(element_cov, self.max_sh)

        gmm_log_probs, d_mean_trans = self.phi_d(element_inv).split(self.num_gaussians, dim=-1)
        half = (self.max_distance - self.min_distance) / 2
        center = (self.min_distance + self.max_distance) / 2
        distance_mean = torch.tanh(d_mean_trans) * half + center
        distance_dist = GMMRef(gmm_log_probs, distance_mean, torch.exp(self.distance_log_stds).clamp(1e-6))
        distance = actions[:, 2:3]

        d = distance.view(B, 1, 1, 1).expand(-1, ce, 1, -1)
        d = torch.cat([d, torch.zeros_like(d)], dim=-1)  # (B, ce, 1, 2) = d + 0i
        cond_cov = self.cg_mix(element_cov, so3.SO3Vec([d]))
        so3_dist = SO3DistRef(cond_cov, self.cg, self.max_sh, self.beta, data['empty'])
        orientation = actions[..., 3:6]

        logps = [
            focus_dist.log_prob(focus.squeeze(-1)),
            element_dist.log_prob(element.squeeze(-1)),
            distance_dist.log_prob(distance.squeeze(-1)),
            so3_dist.log_prob(orientation),
        ]
        log_prob = torch.stack(logps, dim=-1).sum(dim=-1)
        entropy = focus_dist.entropy() + element_dist.entropy()

        trans = self.phi_trans(invariats)
        value_feats = torch.einsum('ba,baf->bf', data['value_mask'].to(dtype), trans)
        value = self.phi_v(value_feats).squeeze(-1)

        out = {'a': actions, 'logp': log_prob, 'ent': entropy, 'v': value}
        if return_internals:
            out.update(covariats=covariats, invariats=invariats, focus_logits=focus_logits,
                       element_logits=element_logits, logps=logps, cond_cov=cond_cov, data=data,
                       ent_parts=[focus_dist.entropy(), element_dist.entropy()],
                       log_z=getattr(so3_dist, 'log_z', None), element_cov=element_cov)
        return out
