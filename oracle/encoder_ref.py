"""Cormorant encoder of the oracle (test infrastructure, see oracle/__init__.py).

Restates, layer by layer, what /root/reference/molgym/agents/covariant/modules.py
wires together (Cormorant.__init__ :11-95, forward :97-114, prepare_input
:116-135, CormorantMixer :138-190) out of risilab/cormorant @ 6a4b6370 (absent
from disk; PARITY UNPINNED).  Submodule / parameter names follow
the attribute names visible at those call sites so that ``state_dict()`` keys
line up with a reference checkpoint's.
"""
import math

import torch
from torch import nn

from . import so3


class RadPolyTrig(nn.Module):
    """32 radial features sin(2 pi s_t r + phi_t) * r^-p (t < 8, p < 4; index
    t*4 + p), zero on masked / zero-length edges, then one Linear(32 -> 2c) per l
    viewed as c complex channels."""

    def __init__(self, max_sh, basis_set, num_channels):
        super().__init__()
        trig, self.rpow = basis_set
        self.max_sh = max_sh
        self.num_channels = num_channels
        ar = torch.arange(trig + 1, dtype=torch.get_default_dtype())
        self.scales = nn.Parameter(torch.cat([ar, ar]).view(1, 1, 1, -1))
        self.phases = nn.Parameter(torch.cat([torch.zeros(trig + 1), math.pi / 2 * torch.ones(trig + 1)]).view(
            1, 1, 1, -1))
        self.num_feat = 2 * (trig + 1) * (self.rpow + 1)
        self.linear = nn.ModuleList([nn.Linear(self.num_feat, 2 * num_channels) for _ in range(max_sh + 1)])

    def features(self, norms, edge_mask):
        mask = (edge_mask * (norms > 0)).unsqueeze(-1)
        r = norms.unsqueeze(-1)
        zero = torch.zeros((), dtype=norms.dtype)
        safe_r = torch.where(mask, r, torch.ones_like(r))
        powers = torch.stack([torch.where(mask, safe_r.pow(-p), zero) for p in range(self.rpow + 1)], dim=-1)
        trig = torch.where(mask, torch.sin(2 * math.pi * self.scales * r + self.phases), zero).unsqueeze(-1)
        return (powers * trig).reshape(norms.shape + (self.num_feat, ))

    def forward(self, norms, edge_mask):
        feats = self.features(norms, edge_mask)
        return so3.SO3Scalar([lin(feats).reshape(norms.shape + (self.num_channels, 2)) for lin in self.linear])


class RadialFilters(nn.Module):
    def __init__(self, max_sh, basis_set, num_channels_out, num_levels):
        super().__init__()
        self.rad_funcs = nn.ModuleList(
            [RadPolyTrig(max_sh[k], basis_set, num_channels_out[k]) for k in range(num_levels)])
        self.tau = [[num_channels_out[k]] * (max_sh[k] + 1) for k in range(num_levels)]

    def forward(self, norms, base_mask):
        return [f(norms, base_mask) for f in self.rad_funcs]


class InputLinear(nn.Module):
    def __init__(self, channels_in, channels_out):
        super().__init__()
        self.channels_out = channels_out
        self.lin = nn.Linear(channels_in, 2 * channels_out)
        self.tau = [channels_out]

    def forward(self, atom_features, atom_mask):
        out = torch.where(atom_mask.unsqueeze(-1), self.lin(atom_features),
                          torch.zeros((), dtype=atom_features.dtype))
        return so3.SO3Vec([out.reshape(atom_features.shape[:2] + (self.channels_out, 1, 2))])


class MixReps(nn.Module):
    """weights[l]: (tau_out[l], tau_in[l], 2); 'rand' init U(-1,1)*gain/max(dims)."""

    def __init__(self, tau_in, num_out, gain):
        super().__init__()
        self.tau_in = list(tau_in)
        self.tau = [num_out] * len(self.tau_in)
        self.weights = nn.ParameterList([nn.Parameter(w) for w in so3.init_mix_weights(self.tau_in, self.tau, gain)])


class CatMixReps(MixReps):
    def __init__(self, taus_in, num_out, gain=1.0):
        super().__init__(so3.cat_tau(taus_in), num_out, gain)

    def forward(self, reps):
        return so3.mix_vec(list(self.weights), so3.cat_reps(reps, cdim=-3))


class CatMixScalar(MixReps):
    def __init__(self, taus_in, num_out, gain=1.0):
        super().__init__(so3.cat_tau(taus_in), num_out, gain)

    def forward(self, scalars):
        return so3.mix_scalar(list(self.weights), so3.cat_reps(scalars, cdim=-2))


class DotMatrix(nn.Module):
    """(psi_i . psi_j)_c = sum_m (-1)^m psi_i[c, m] psi_j[c, -m] (complex, no
    conjugation) for every part l; parts concatenated along channels and the
    same tensor repeated once per input part (cat=True)."""

    def __init__(self, tau_in):
        super().__init__()
        self.tau = [sum(tau_in)] * len(tau_in)

    def forward(self, reps):
        dots = []
        for l, part in zip(reps.ells, reps):
            sign = torch.tensor([(-1.0) ** m for m in range(-l, l + 1)], dtype=part.dtype).view(-1, 1)
            a = part.unsqueeze(-4)  # (B, N, 1, c, m, 2)  atom i
            b = (part.flip(-2) * sign).unsqueeze(-5)  # (B, 1, N, c, m, 2)  atom j
            dots.append(so3.cmul(a, b).sum(dim=-2))  # (B, N, N, c, 2)
        cat = torch.cat(dots, dim=-2)
        return so3.SO3Scalar([cat] * len(reps))


class MaskLevel(nn.Module):
    """cutoff_type ['soft']: edge *= edge_mask * sigmoid((rad - r) / width) with
    fixed (non-learnable) per-channel rad / width."""

    def __init__(self, num_channels, soft_cut_rad, soft_cut_width, eps=1e-3):
        super().__init__()
        self.rad = float(max(eps, abs(soft_cut_rad)))
        self.width = float(max(eps, abs(soft_cut_width)))

    def forward(self, edge_net, edge_mask, norms):
        m = edge_mask.to(norms.dtype) * torch.sigmoid((self.rad - norms) / self.width)
        m = m.unsqueeze(-1).unsqueeze(-1)
        return so3.SO3Scalar([p * m for p in edge_net])


class EdgeLevel(nn.Module):
    def __init__(self, tau_atom, tau_edge, tau_pos, nout, soft_cut_rad, soft_cut_width):
        super().__init__()
        self.dot_matrix = DotMatrix(tau_atom)
        self.cat_mix = CatMixScalar([tau_edge, self.dot_matrix.tau, tau_pos], nout, gain=1.0)
        self.tau = self.cat_mix.tau
        self.mask_layer = MaskLevel(nout, soft_cut_rad, soft_cut_width)

    def forward(self, edge_in, atom_reps, pos_funcs, base_mask, norms):
        mixed = self.cat_mix([edge_in, self.dot_matrix(atom_reps), pos_funcs])
        return self.mask_layer(mixed, base_mask, norms)


class AtomLevel(nn.Module):
    def __init__(self, tau_in, tau_pos, maxl, num_channels, level_gain, cg):
        super().__init__()
        self.maxl, self.cg = maxl, cg
        tau_ag = so3.cg_product_tau(tau_pos, tau_in, maxl)
        tau_sq = so3.cg_product_tau(tau_in, tau_in, maxl)
        self.cat_mix = CatMixReps([tau_ag, list(tau_in), tau_sq], num_channels, gain=level_gain)
        self.tau = self.cat_mix.tau

    def forward(self, atom_reps, edge_reps):
        cg = self.cg.to(atom_reps[0].dtype)
        ag = so3.cg_product(cg, edge_reps, atom_reps, self.maxl, aggregate=True)
        sq = so3.cg_product(cg, atom_reps, atom_reps, self.maxl)
        return self.cat_mix([ag, atom_reps, sq])


class CormorantCG(nn.Module):
    def __init__(self, maxl, max_sh, tau_in_atom, tau_pos, num_cg_levels, num_channels, level_gain, soft_cut_rad,
                 soft_cut_width, cg):
        super().__init__()
        tau_atom, tau_edge = tau_in_atom, None
        self.edge_levels, self.atom_levels = nn.ModuleList(), nn.ModuleList()
        for k in range(num_cg_levels):
            e = EdgeLevel(tau_atom, tau_edge, tau_pos[k], num_channels[k], soft_cut_rad[k], soft_cut_width[k])
            self.edge_levels.append(e)
            tau_edge = e.tau
            a = AtomLevel(tau_atom, tau_edge, maxl[k], num_channels[k + 1], level_gain[k], cg)
            self.atom_levels.append(a)
            tau_atom = a.tau

    def forward(self, atom_reps, edge_net, edge_mask, rad_funcs, norms, sph_harm):
        atoms_all, edges_all = [], []
        for k, (atom_level, edge_level) in enumerate(zip(self.atom_levels, self.edge_levels)):
            edge_net = edge_level(edge_net, atom_reps, rad_funcs[k], edge_mask, norms)
            edge_reps = so3.scalar_times_vec(edge_net, sph_harm)
            atom_reps = atom_level(atom_reps, edge_reps)
            atoms_all.append(atom_reps)
            edges_all.append(edge_net)
        return atoms_all, edges_all


class Cormorant(nn.Module):
    """Encoder: data dict -> last-level SO3Vec, list over l of (B, N, Co, 2l+1, 2)."""

    def __init__(self, maxl, max_sh, num_cg_levels, num_channels, num_species, soft_cut_rad, soft_cut_width,
                 level_gain, charge_power, basis_set, charge_scale, bag_scale, cg):
        super().__init__()
        n = num_cg_levels
        self.maxl_list, self.max_sh = [maxl] * n, [max_sh] * n
        self.charge_power, self.charge_scale, self.bag_scale = charge_power, charge_scale, bag_scale
        self.cg = cg
        self.rad_funcs = RadialFilters(self.max_sh, basis_set, num_channels, n)
        num_scalars_in = num_species * (charge_power + 1) + num_species
        self.input_func_atom = InputLinear(num_scalars_in, num_channels[0])
        self.cormorant_cg = CormorantCG(self.maxl_list, self.max_sh, self.input_func_atom.tau, self.rad_funcs.tau, n,
                                        num_channels, [level_gain] * n, [soft_cut_rad] * n, [soft_cut_width] * n, cg)

    def prepare_input(self, data):
        dtype = data['positions'].dtype
        one_hot = data['one_hot'].to(dtype)
        charges = data['charges'].to(dtype)
        powers = torch.arange(self.charge_power + 1, dtype=dtype)
        ct = (charges.unsqueeze(-1) / self.charge_scale).pow(powers)  # (B, N, P)
        ct = (one_hot.unsqueeze(-1) * ct.unsqueeze(-2)).reshape(charges.shape[:2] + (-1, ))
        bag = (data['bags'] / self.bag_scale).unsqueeze(1).expand(ct.shape[:-1] + (-1, ))
        return torch.cat([ct, bag], dim=-1)

    def forward(self, data, return_all=False):
        pos = data['positions']
        cg = self.cg.to(pos.dtype)
        atom_scalars = self.prepare_input(data)
        sph, norms = so3.spherical_harmonics_rel(cg, pos, pos, max(self.max_sh), conj=True)
        rad = self.rad_funcs(norms, data['edge_mask'] * (norms > 0))
        atom_in = self.input_func_atom(atom_scalars, data['atom_mask'])
        atoms_all, edges_all = self.cormorant_cg(atom_in, None, data['edge_mask'], rad, norms, sph)
        if return_all:
            return atoms_all, edges_all, dict(sph=sph, norms=norms, rad=rad, atom_in=atom_in,
                                              atom_scalars=atom_scalars)
        return atoms_all[-1]


class CormorantMixer(nn.Module):
    """ag = other (x) in; sq = ag (x) ag; mix(cat[ag, sq, in]) -> num_channels."""

    def __init__(self, tau_in, tau_other, maxl, num_channels, level_gain, cg):
        super().__init__()
        self.maxl, self.cg = maxl, cg
        tau_ag = so3.cg_product_tau(tau_other, tau_in, maxl)
        tau_sq = so3.cg_product_tau(tau_ag, tau_ag, maxl)
        self.cat_mix = CatMixReps([tau_ag, tau_sq, list(tau_in)], num_channels, gain=level_gain)

    def forward(self, atom_reps, other_reps):
        cg = self.cg.to(atom_reps[0].dtype)
        ag = so3.cg_product(cg, other_reps, atom_reps, self.maxl)
        sq = so3.cg_product(cg, ag, ag, self.maxl)
        return self.cat_mix([ag, sq, atom_reps])
