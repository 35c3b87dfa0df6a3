"""SO(3) primitives of the oracle (test infrastructure, see oracle/__init__.py).

Restates the published algorithm of risilab/cormorant @ 6a4b6370 (not on disk;
call sites: /root/reference/molgym/agents/covariant/modules.py:4-8,
agent.py:6-7,59,93-98).  PARITY UNPINNED for everything except Y_lm, which the
reference pins at tests/agents/covariant/test_sphs.py:28-32,46-53.

Data model (reference comments at agent.py:216, so3_tools.py:48,109,120):
an SO(3) vector is a list over l of tensors (..., tau_l, 2l+1, 2), last axis
(re, im); an SO(3) scalar is a list over l of tensors (..., tau_l, 2).
"""
import math
from itertools import zip_longest

import torch


# ----------------------------------------------------------------------------
# Clebsch-Gordan coefficients <l1 m1 l2 m2 | l m>, Racah's closed form
# ----------------------------------------------------------------------------
def clebsch_gordan(l1, m1, l2, m2, l, m):
    if m1 + m2 != m or l < abs(l1 - l2) or l > l1 + l2:
        return 0.0
    if abs(m1) > l1 or abs(m2) > l2 or abs(m) > l:
        return 0.0
    f = math.factorial
    pref = (2 * l + 1) * f(l1 + l2 - l) * f(l1 - l2 + l) * f(-l1 + l2 + l) / f(l1 + l2 + l + 1)
    pref *= f(l + m) * f(l - m) * f(l1 - m1) * f(l1 + m1) * f(l2 - m2) * f(l2 + m2)
    total = 0.0
    for k in range(0, l1 + l2 + l + 1):
        d = [k, l1 + l2 - l - k, l1 - m1 - k, l2 + m2 - k, l - l2 + m1 + k, l - l1 - m2 + k]
        if min(d) < 0:
            continue
        den = 1
        for x in d:
            den *= f(x)
        total += (-1.0) ** k / den
    return math.sqrt(pref) * total


class CGTable:
    """(l1, l2) -> matrix [(sum_l 2l+1), (2l1+1)(2l2+1)], rows stacked over
    l = |l1-l2| .. l1+l2, column index m1_idx*(2l2+1) + m2_idx."""

    def __init__(self, maxl, dtype=torch.float32):
        self.maxl = maxl
        self.mats = {}
        for l1 in range(maxl + 1):
            for l2 in range(maxl + 1):
                rows = []
                for l in range(abs(l1 - l2), l1 + l2 + 1):
                    for m in range(-l, l + 1):
                        row = [clebsch_gordan(l1, m1, l2, m2, l, m)
                               for m1 in range(-l1, l1 + 1) for m2 in range(-l2, l2 + 1)]
                        rows.append(row)
                self.mats[(l1, l2)] = torch.tensor(rows, dtype=torch.float64).to(dtype)

    def to(self, dtype):
        out = CGTable.__new__(CGTable)
        out.maxl = self.maxl
        out.mats = {k: v.to(dtype) for k, v in self.mats.items()}
        return out


# ----------------------------------------------------------------------------
# containers
# ----------------------------------------------------------------------------
class SO3Vec(list):
    """list over l of (..., tau, 2l+1, 2)."""

    @property
    def ells(self):
        return [(p.shape[-2] - 1) // 2 for p in self]

    @property
    def tau(self):
        return [p.shape[-3] for p in self]


class SO3Scalar(list):
    """list over l of (..., tau, 2)."""

    @property
    def tau(self):
        return [p.shape[-2] for p in self]


def cmul(a, b):
    """complex product on a trailing (re, im) axis, with broadcasting."""
    ar, ai = a.unbind(-1)
    br, bi = b.unbind(-1)
    return torch.stack([ar * br - ai * bi, ar * bi + ai * br], dim=-1)


def scalar_times_vec(scal, vec):
    """SO3Scalar * SO3Vec: channel-wise complex scale of every m component."""
    return SO3Vec([cmul(s.unsqueeze(-2), v) for s, v in zip(scal, vec)])


# ----------------------------------------------------------------------------
# CG product (channel-wise), optionally aggregating over the second atom index
# ----------------------------------------------------------------------------
def _kron(z1, z2, aggregate):
    """z1 (..., tau, n1, 2) x z2 (..., tau, n2, 2) -> (..., tau, n1*n2, 2).
    aggregate: z1 is (B, N, N, tau, n1, 2), z2 is (B, N, tau, n2, 2) and the
    product is summed over the second atom index j."""
    n1, n2 = z1.shape[-2], z2.shape[-2]
    a = z1.unsqueeze(-2)  # (..., tau, n1, 1, 2)
    b = z2.unsqueeze(-3)  # (..., tau, 1, n2, 2)
    if aggregate:
        b = b.unsqueeze(1)  # (B, 1, N, tau, 1, n2, 2)
        prod = cmul(a, b).sum(dim=2)
    else:
        prod = cmul(a, b)
    return prod.reshape(prod.shape[:-3] + (n1 * n2, 2))


def cg_product(cg, rep1, rep2, maxl, aggregate=False):
    """For every (l1, l2) (l1 outer, l2 inner) and every l in
    [|l1-l2|, min(l1+l2, maxl)] append CG-projected kron(rep1_l1, rep2_l2) to
    output part l along the channel axis."""
    out = [[] for _ in range(maxl + 1)]
    for l1, p1 in zip(rep1.ells, rep1):
        for l2, p2 in zip(rep2.ells, rep2):
            lmin, lmax = abs(l1 - l2), min(l1 + l2, maxl)
            if lmin > lmax:
                continue
            nrows = (lmax + 1) ** 2 - lmin ** 2
            mat = cg.mats[(l1, l2)][:nrows]  # (rows, n1*n2)
            kr = _kron(p1, p2, aggregate)  # (..., tau, n1*n2, 2)
            dec = torch.einsum('rk,...tkx->...trx', mat, kr)
            off = 0
            for l in range(lmin, lmax + 1):
                out[l].append(dec[..., off:off + 2 * l + 1, :])
                off += 2 * l + 1
    return SO3Vec([torch.cat(parts, dim=-3) for parts in out if parts])


def cg_product_tau(tau1, tau2, maxl):
    out = [0] * (maxl + 1)
    for l1, t1 in enumerate(tau1):
        for l2, t2 in enumerate(tau2):
            if t1 == 0 or t2 == 0:
                continue
            assert t1 == t2, 'channel-wise CG product needs equal channel counts'
            for l in range(abs(l1 - l2), min(l1 + l2, maxl) + 1):
                out[l] += t1
    while out and out[-1] == 0:
        out.pop()
    return out


# ----------------------------------------------------------------------------
# spherical harmonics by the CG recursion Y_l ~ (Y_{l-1} x Y_1)_l
# ----------------------------------------------------------------------------
def spherical_harmonics(cg, pos, maxl, normalize=True, conj=False, sh_norm='unit'):
    """pos (..., 3) -> SO3Vec of (..., 1, 2l+1, 2).  sh_norm 'qm' is the standard
    Condon-Shortley Y_l^m; 'unit' multiplies part l by sqrt(4 pi / (2l+1))."""
    shape = pos.shape[:-1]
    pos = pos.reshape(-1, 3)
    if normalize:
        norm = pos.norm(dim=-1, keepdim=True)
        pos = torch.where(norm > 0, pos / norm, torch.zeros_like(pos))
    x, y, z = pos.unbind(-1)
    zero = torch.zeros_like(x)
    sgn = 1.0 if conj else -1.0
    y1 = torch.stack([
        torch.stack([x, sgn * y], -1) / math.sqrt(2.0),
        torch.stack([z, zero], -1),
        torch.stack([-x, sgn * y], -1) / math.sqrt(2.0),
    ], dim=-2).unsqueeze(-3) * math.sqrt(3.0 / (4 * math.pi))  # (P, 1, 3, 2)
    y0 = torch.zeros(pos.shape[0], 1, 1, 2, dtype=pos.dtype, device=pos.device)
    y0[..., 0] = math.sqrt(1.0 / (4 * math.pi))
    parts = [y0]
    if maxl >= 1:
        parts.append(y1)
    cur = y1
    for l in range(2, maxl + 1):
        cur = cg_product(cg, SO3Vec([cur]), SO3Vec([y1]), maxl=l)[-1]
        c0 = clebsch_gordan(l - 1, 0, 1, 0, l, 0)
        cur = cur * (math.sqrt(4 * math.pi * (2 * l + 1) / (3.0 * (2 * l - 1))) / c0)
        parts.append(cur)
    if sh_norm == 'unit':
        parts = [p * math.sqrt(4 * math.pi / (2 * l + 1)) for l, p in enumerate(parts)]
    elif sh_norm != 'qm':
        raise ValueError(sh_norm)
    return SO3Vec([p.reshape(shape + (1, 2 * l + 1, 2)) for l, p in enumerate(parts)])


def spherical_harmonics_rel(cg, pos1, pos2, maxl, conj=False, sh_norm='unit'):
    """Y of all x_i - x_j, plus the pair distances (B, N, N)."""
    rel = pos1.unsqueeze(-2) - pos2.unsqueeze(-3)
    return spherical_harmonics(cg, rel, maxl, True, conj, sh_norm), rel.norm(dim=-1)


# ----------------------------------------------------------------------------
# concatenate / mix
# ----------------------------------------------------------------------------
def cat_reps(reps_list, cdim):
    """Per-l concatenation along the channel axis; parts absent from a rep are
    skipped for that l (zip_longest semantics)."""
    reps_list = [r for r in reps_list if r is not None]
    out = []
    for parts in zip_longest(*reps_list, fillvalue=None):
        parts = [p for p in parts if p is not None]
        out.append(torch.cat(parts, dim=cdim))
    return out


def cat_tau(taus):
    taus = [t for t in taus if t]
    n = max(len(t) for t in taus)
    return [sum(t[l] for t in taus if l < len(t)) for l in range(n)]


def mix_vec(weights, vec):
    """weights[l] (t_out, t_in, 2) applied to vec[l] (..., t_in, m, 2)."""
    out = []
    for w, p in zip(weights, vec):
        wr, wi = w.unbind(-1)
        pr, pi = p.unbind(-1)
        out.append(torch.stack([wr @ pr - wi @ pi, wi @ pr + wr @ pi], dim=-1))
    return SO3Vec(out)


def mix_scalar(weights, scal):
    """weights[l] (t_out, t_in, 2) applied to scal[l] (..., t_in, 2)."""
    out = []
    for w, p in zip(weights, scal):
        wr, wi = w.unbind(-1)
        pr, pi = p.unbind(-1)
        re = torch.einsum('oi,...i->...o', wr, pr) - torch.einsum('oi,...i->...o', wi, pi)
        im = torch.einsum('oi,...i->...o', wi, pr) + torch.einsum('oi,...i->...o', wr, pi)
        out.append(torch.stack([re, im], dim=-1))
    return SO3Scalar(out)


def init_mix_weights(tau_in, tau_out, gain, generator=None):
    """'rand' init: U(-1, 1) per (re, im) entry, scaled by gain / max(dims)."""
    ws = []
    for ti, to in zip(tau_in, tau_out):
        w = 2 * torch.rand(to, ti, 2, generator=generator) - 1
        ws.append(w * (gain / max(to, ti, 2)))
    return ws
