"""SchNetAC.step of the oracle (test infrastructure, see oracle/__init__.py).

Follows /root/reference/molgym/agents/internal/agent.py:17-353 and internal/zmat.py:6-133.  The SchNet
embedding is schnetpack 0.3 (requirements.txt:24; absent from disk, PARITY UNPINNED): representation.SchNet
(n_atom_basis=64) with its defaults n_filters=128, n_interactions=3, cutoff=5.0, n_gaussians=25, max_z=100,
cosine cutoff, shifted softplus, all other atoms as neighbours (SimpleEnvironmentProvider via AtomsConverter).
zmat is importable in the build container and pinned by tests/golden/g8_zmat.npz.
"""
import math
from typing import List

import numpy as np
import torch
from torch import nn

from .covariant_ref import MLP, CategoricalRef, masked_softmax, to_one_hot


def ssp(x):
    return torch.nn.functional.softplus(x) - math.log(2.0)


class Dense(nn.Linear):
    def __init__(self, n_in, n_out, bias=True, activation=None):
        super().__init__(n_in, n_out, bias)
        nn.init.xavier_uniform_(self.weight)
        if bias:
            nn.init.zeros_(self.bias)
        self.activation = activation

    def forward(self, x):
        y = super().forward(x)
        return self.activation(y) if self.activation else y


class SchNetInteraction(nn.Module):
    def __init__(self, n_atom_basis, n_spatial_basis, n_filters, cutoff):
        super().__init__()
        self.cutoff = cutoff
        self.filter_network = nn.Sequential(Dense(n_spatial_basis, n_filters, activation=ssp),
                                            Dense(n_filters, n_filters))
        self.cfconv = nn.Module()  # schnetpack's CFConv owns in2f / f2out
        self.cfconv.in2f = Dense(n_atom_basis, n_filters, bias=False)
        self.cfconv.f2out = Dense(n_filters, n_atom_basis, activation=ssp)
        self.dense = Dense(n_atom_basis, n_atom_basis)

    def forward(self, x, r_ij, f_ij):
        """one molecule: x (n, F), r_ij (n, n) with zero diagonal, f_ij (n, n, G); neighbours = all j != i."""
        n = x.shape[0]
        w = self.filter_network(f_ij)
        c = 0.5 * (torch.cos(r_ij * math.pi / self.cutoff) + 1.0) * (r_ij < self.cutoff).to(x.dtype)
        w = w * c.unsqueeze(-1)
        y = self.cfconv.in2f(x)  # (n, n_filters)
        mask = (1.0 - torch.eye(n, dtype=x.dtype)).unsqueeze(-1)
        agg = (y.unsqueeze(0) * w * mask).sum(dim=1)
        return self.dense(self.cfconv.f2out(agg))


class SchNet(nn.Module):
    def __init__(self, n_atom_basis=64, n_filters=128, n_interactions=3, cutoff=5.0, n_gaussians=25, max_z=100):
        super().__init__()
        self.embedding = nn.Embedding(max_z, n_atom_basis, padding_idx=0)
        self.register_buffer('offsets', torch.linspace(0.0, cutoff, n_gaussians), persistent=False)
        self.coeff = -0.5 / float((cutoff / (n_gaussians - 1))**2)
        self.interactions = nn.ModuleList(
            [SchNetInteraction(n_atom_basis, n_gaussians, n_filters, cutoff) for _ in range(n_interactions)])

    def forward(self, numbers, positions):
        """numbers (n,) long, positions (n, 3) -> (n, n_atom_basis)"""
        x = self.embedding(numbers)
        r_ij = (positions.unsqueeze(0) - positions.unsqueeze(1)).norm(dim=-1)
        r_ij = r_ij * (1.0 - torch.eye(len(numbers), dtype=r_ij.dtype))
        f_ij = torch.exp(self.coeff * (r_ij.unsqueeze(-1) - self.offsets.to(r_ij.dtype))**2)
        for inter in self.interactions:
            x = x + inter(x, r_ij, f_ij)
        return x


# ---- z-matrix internal coordinates (internal/zmat.py:6-63), float64 numpy ------------------------------------
def get_distance(p_i, p_j):
    """zmat.py:6-14 (also the sort key of the placement, zmat.py:113)"""
    return np.sqrt(np.sum(np.square(p_i - p_j)))


def get_angle(p_i, p_j, p_k):
    """zmat.py:17-31: angle at j, from |a x b| and a . b"""
    a, b = p_i - p_j, p_k - p_j
    return np.arctan2(np.linalg.norm(np.cross(a, b)), np.dot(a, b))


def get_dihedral(p_i, p_j, p_k, p_l):
    """zmat.py:34-63: signed dihedral i-j-k-l in (-pi, pi]; NaN for collinear points (the normals have zero length), as the
    reference's test_dihedral_nan expects"""
    b0, b1, b2 = p_j - p_i, p_k - p_j, p_l - p_k
    n1 = np.cross(b0, b1)
    n1 = n1 / np.linalg.norm(n1)
    n2 = np.cross(b2, b1)
    n2 = n2 / np.linalg.norm(n2)
    m1 = np.cross(n1, b1) / np.linalg.norm(b1)
    psi = np.arctan2(np.dot(m1, n2), np.dot(n1, n2))
    return -psi - np.pi if psi < 0 else np.pi - psi


# ---- z-matrix placement (internal/zmat.py:66-133), float64 numpy --------------------------------------------
def position_point(p0, p1, p2, distance, angle, dihedral):
    x = distance * np.cos(angle)
    y = distance * np.cos(dihedral) * np.sin(angle)
    z = distance * np.sin(dihedral) * np.sin(angle)
    v_a = p1 - p0
    v_b = (p2 - p1) / np.linalg.norm(p2 - p1)
    c_ab = np.cross(v_a, v_b)
    c_ab = c_ab / np.linalg.norm(c_ab)
    c_ab_b = np.cross(c_ab, v_b)
    return p2 - v_b * x + c_ab_b * y + c_ab * z


def position_atom(positions: List[np.ndarray], focus: int, distance, angle, dihedral):
    if focus > len(positions):
        raise RuntimeError('Focus greater than number of atoms')
    if len(positions) == 0:
        return np.zeros(3)
    f = positions[focus]
    order = sorted(positions, key=lambda p: np.sqrt(np.sum(np.square(p - f))))
    aux1, aux0 = np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0])
    if len(positions) == 1:
        p2 = order[0]
        p1, p0 = p2 + aux1, p2 + aux0
    elif len(positions) == 2:
        p2, p1 = order[0], order[1]
        p0 = p2 + p1 + aux0 + aux1
    else:
        p2, p1, p0 = order[0], order[1], order[2]
    return position_point(p0, p1, p2, distance, angle, dihedral)


class SchNetACRef(nn.Module):
    def __init__(self, zs: List[int], canvas_size: int, min_max_distance, network_width: int):
        super().__init__()
        self.zs, self.num_atoms, self.num_zs = list(zs), canvas_size, len(zs)
        self.num_afeats, self.num_latent_beta = network_width // 2, network_width // 4
        self.num_latent = self.num_afeats + self.num_latent_beta
        self.embedding_fn = SchNet(n_atom_basis=self.num_afeats)
        w = network_width
        self.phi_beta = MLP(self.num_zs, (w, self.num_latent_beta))
        self.phi_focus = MLP(self.num_latent, (w, 1))
        self.phi_element = MLP(self.num_latent, (w, self.num_zs))
        self.phi_continuous = MLP(self.num_latent + self.num_zs, (w, 3))
        self.phi_kappa = MLP(self.num_latent, (w, 1))
        self.log_stds = nn.Parameter(torch.log(torch.tensor([0.15, 0.25, 0.25])))
        lo, hi = min_max_distance
        self.action_width = [hi - lo, math.pi, math.pi]
        self.action_center = [0.5 * (hi + lo), 0.5 * math.pi, 0.5 * math.pi]
        self.critic = MLP(self.num_latent, (w, w, 1))

    def _atoms(self, observation):
        canvas, bag = observation
        return [(self.zs[l], np.asarray(x, dtype=np.float64)) for l, x in canvas if self.zs[l] != 0], bag

    def _embed(self, numbers, positions, dtype):
        return self.embedding_fn(torch.tensor(numbers, dtype=torch.long),
                                 torch.tensor(np.asarray(positions), dtype=torch.float32).to(dtype))

    def step(self, observations, actions, dtype=torch.float32):
        """action evaluation (actions (B, 7): stop, focus, element, distance, angle, dihedral, kappa)."""
        B, N = len(observations), self.num_atoms
        actions = torch.as_tensor(np.asarray(actions), dtype=torch.float32).to(dtype)
        feats = torch.zeros(B, N, self.num_afeats, dtype=dtype)
        focus_mask = torch.zeros(B, N, dtype=torch.bool)
        element_count = torch.zeros(B, self.num_zs, dtype=dtype)
        action_mask = torch.zeros(B, 6, dtype=dtype)
        parsed = [self._atoms(o) for o in observations]
        rows = []
        for i, (atoms, bag) in enumerate(parsed):
            n = len(atoms)
            if n:
                rows.append(self._embed([z for z, _ in atoms], [p for _, p in atoms], dtype))
                focus_mask[i, :n] = True
            else:
                rows.append(torch.zeros(0, self.num_afeats, dtype=dtype))
                focus_mask[i, :1] = True
            element_count[i] = torch.tensor(bag, dtype=dtype)
            action_mask[i] = torch.tensor([n >= 1, 1.0, n >= 1, n >= 2, n >= 3, n >= 3], dtype=dtype)
        feats = torch.stack([torch.cat([r, torch.zeros(N - r.shape[0], self.num_afeats, dtype=dtype)]) for r in rows])
        element_mask = element_count > 0
        latent_bag = self.phi_beta(element_count)
        latent = torch.cat([feats, latent_bag.unsqueeze(1).expand(-1, N, -1)], dim=-1)
        focus_logits = self.phi_focus(latent).squeeze(-1)
        focus_dist = CategoricalRef(masked_softmax(focus_logits, focus_mask))
        focus = torch.round(actions[:, 1:2]).long()
        focus_oh = to_one_hot(focus, N).to(dtype)
        focused = torch.einsum('ba,baf->bf', focus_oh, latent)
        element_logits = self.phi_element(focused)
        element_dist = CategoricalRef(masked_softmax(element_logits, element_mask))
        element = torch.round(actions[:, 2:3]).long()
        element_oh = to_one_hot(element, self.num_zs).to(dtype)
        means = torch.tanh(self.phi_continuous(torch.cat([focused, element_oh], dim=-1)))
        cont_lp, cont_vals = [], []
        for k in range(3):
            mean = means[:, k:k + 1] * self.action_width[k] / 2 + self.action_center[k]
            scale = torch.exp(1e-6 + self.log_stds[k])
            val = actions[:, 3 + k:4 + k]
            cont_lp.append(-((val - mean)**2) / (2 * scale**2) - torch.log(scale) - math.log(math.sqrt(2 * math.pi)))
            cont_vals.append(val)
        latent_bag_next = self.phi_beta(element_count - element_oh)
        v_kappa = []
        for sign in (1.0, -1.0):
            rows = []
            for i, (atoms, bag) in enumerate(parsed):
                new = position_atom([p for _, p in atoms], int(round(float(actions[i, 1]))), float(actions[i, 3]),
                                    float(actions[i, 4]), sign * float(actions[i, 5]))
                z_new = self.zs[int(round(float(actions[i, 2])))]
                emb = self._embed([z for z, _ in atoms] + [z_new], [p for _, p in atoms] + [new], dtype)
                rows.append(emb[-1])
            v_kappa.append(self.phi_kappa(torch.cat([torch.stack(rows), latent_bag_next], dim=-1)))
        kappa_logits = torch.cat(v_kappa, dim=-1)
        kappa_logp = kappa_logits - kappa_logits.logsumexp(-1, keepdim=True)
        kappa = torch.round(actions[:, 6]).long()
        sum_feats = torch.einsum('ba,baf->bf', focus_mask.to(dtype), feats)
        v = self.critic(torch.cat([sum_feats, latent_bag], dim=-1)).squeeze(-1)
        log_prob = torch.cat([focus_dist.log_prob(focus.squeeze(-1)).unsqueeze(-1),
                              element_dist.log_prob(element.squeeze(-1)).unsqueeze(-1)] + cont_lp +
                             [kappa_logp.gather(-1, kappa.unsqueeze(-1))], dim=-1) * action_mask
        ent = torch.stack([focus_dist.entropy(), element_dist.entropy()], dim=-1) * action_mask[:, 0:2]
        return {'a': actions, 'logp': log_prob.sum(-1), 'ent': ent.sum(-1), 'v': v}
