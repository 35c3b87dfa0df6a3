"""The `dists` entry of CovariantAC.step()'s return value.

The reference returns ``[focus_dist, element_dist, distance_dist, so3_dist]``
(/root/reference/molgym/agents/covariant/agent.py:325-331): two
``torch.distributions.Categorical(probs=masked_softmax(...))`` (:224-226, :243-245), a
``GaussianMixtureModel`` (gmm.py:8-27) and an ``ExpSO3Distribution`` / ``SO3Distribution``
(spherical_dists.py:79-286).  The PPO loss never touches them (ppo.py:33-48 reads logp / ent / v),
the reference's tests and analysis scripts do (tests/agents/covariant/test_agent.py:47,58,86:
``dists[-1].coefficients.apply_wigner(D)``, ``dists[-1].log_prob(grid)``).

Here the list is LAZY: `step()` hands back a `StepDists` holding the workspace of the forward that
just ran; the distribution objects are assembled from the head outputs the kernels left there
(focus / element logits, GMM parameters, conditioned orientation coefficients, log Z) only when an
element is indexed, so the training path pays nothing for them.  The two Categoricals and the mixture
are the same torch classes the reference builds (tiny tensors); the orientation density is evaluated
by `mg_so3_density` (csrc/dists.inc) on whatever points the caller brings.
"""
import ctypes as C
import math
from collections.abc import Sequence
from typing import List, Optional

import numpy as np
import torch
import torch.distributions as D

from .. import _lib


class SO3VecLite(list):
    """list over l of (..., tau, 2l+1, 2) tensors -- the surface of cormorant.so3_lib.SO3Vec the
    reference touches on distribution coefficients (`.ells`, iteration, indexing, `apply_wigner`)."""

    @property
    def ells(self) -> List[int]:
        return [(p.shape[-2] - 1) // 2 for p in self]

    @property
    def maxl(self) -> int:
        return len(self) - 1

    def apply_wigner(self, wigner_d, dir: str = 'left') -> 'SO3VecLite':
        """Rotate every part with the Wigner matrices `wigner_d` (list over l of (2l+1, 2l+1, 2) real/imag
        tensors, or complex (2l+1, 2l+1) tensors): out[..., t, m] = sum_m' D[m, m'] part[..., t, m']
        ('left'; 'right' contracts the first index of D instead)."""
        out = []
        for part, d in zip(self, wigner_d):
            d = torch.as_tensor(d)
            dc = d if d.is_complex() else torch.complex(d[..., 0], d[..., 1])
            dc = dc.to(device=part.device)
            z = torch.complex(part[..., 0], part[..., 1]).to(dc.dtype)
            eq = '...m,nm->...n' if dir == 'left' else '...m,mn->...n'
            r = torch.einsum(eq, z, dc)
            out.append(torch.stack([r.real, r.imag], dim=-1).to(part.dtype))
        return SO3VecLite(out)


def masked_softmax(logits: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """molgym/modules.py:26-27: torch-scatter scatter_softmax keyed by the mask (two groups per row, eps 1e-12 in
    the denominator), times the mask."""
    out = torch.zeros_like(logits)
    for sel in (mask, ~mask):
        grp = torch.where(sel, logits, torch.full_like(logits, -math.inf))
        mx = grp.max(dim=-1, keepdim=True).values
        mx = torch.where(torch.isfinite(mx), mx, torch.zeros_like(mx))
        ex = torch.where(sel, (logits - mx).exp(), torch.zeros_like(logits))
        out = out + ex / (ex.sum(dim=-1, keepdim=True) + 1e-12)
    return out * mask


class GaussianMixtureModel(D.MixtureSameFamily):
    """gmm.py:8-27 (same construction, same argmax-by-sampling)."""

    def __init__(self, log_probs: torch.Tensor, means: torch.Tensor, stds: torch.Tensor, validate_args=None):
        super().__init__(mixture_distribution=D.Categorical(logits=log_probs, validate_args=validate_args),
                         component_distribution=D.Normal(loc=means, scale=stds, validate_args=validate_args),
                         validate_args=validate_args)

    def argmax(self, count: int = 128) -> torch.Tensor:
        samples = self.sample(torch.Size((count, )))
        best = torch.argmax(self.log_prob(samples), dim=0).unsqueeze(0)
        return torch.gather(samples, dim=0, index=best).squeeze(0)


def fibonacci_grid(n: int) -> np.ndarray:
    """so3_tools.generate_fibonacci_grid (so3_tools.py:8-19)."""
    i = np.arange(n)
    theta, phi = np.arccos(1 - 2 * (i + 0.5) / n), 2 * np.pi * i / ((1 + 5**0.5) / 2)
    return np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], axis=-1)


class SO3DensityHIP(D.Distribution):
    """ExpSO3Distribution (beta given) / SO3Distribution (beta None) with the coefficients the heads kernel
    produced; density evaluation on the device (mg_so3_density)."""
    arg_constraints = {}  # type: ignore
    has_rsample = False

    def __init__(self, cond: SO3VecLite, beta: Optional[float], log_z: Optional[torch.Tensor],
                 empty: Optional[torch.Tensor]):
        B = cond[0].shape[0]
        super().__init__(torch.Size((B, )), event_shape=torch.Size((3, )), validate_args=False)
        self.device, self.dtype = cond[0].device, cond[0].dtype
        self.beta = beta
        # normalize_alms (so3_tools.py:56-75): k = sum_{l,m,x} (sum_tau a)^2, clamp 1e-10
        k = sum((p.sum(dim=-3)**2).sum(dim=(-1, -2)) for p in cond)
        inv = torch.rsqrt(k.clamp(min=1e-10)).view(B, 1, 1, 1)
        self.coefficients = SO3VecLite([p * inv for p in cond])
        self._summed = torch.cat([p.sum(dim=-3) for p in self.coefficients], dim=-2).contiguous()  # (B, 25, 2)
        self.log_z = log_z
        self.empty = None if beta is not None else empty
        self._empty_u8 = None if self.empty is None else self.empty.to(torch.uint8).contiguous()

    def _eval(self, value: torch.Tensor, mode: int) -> torch.Tensor:
        value = torch.as_tensor(value, dtype=torch.float32, device=self.device)
        B = self._summed.shape[0]
        if value.shape[-1] != 3 or value.dim() < 2 or value.shape[-2] not in (1, B):
            raise RuntimeError(f'value of shape {tuple(value.shape)} does not broadcast against batch shape ({B},) '
                               'with event shape (3,)')
        lead, Bp = value.shape[:-2], value.shape[-2]
        pts = value.reshape(-1, Bp, 3).contiguous()
        S = pts.shape[0]
        out = torch.empty(S, B, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mg_so3_density(
                B, S, Bp, C.c_void_p(self._summed.data_ptr()), C.c_void_p(pts.data_ptr()),
                0 if self.beta is None else 1, 0.0 if self.beta is None else float(self.beta),
                C.c_void_p(0 if self.log_z is None else self.log_z.data_ptr()),
                C.c_void_p(0 if self._empty_u8 is None else self._empty_u8.data_ptr()), mode,
                C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out.reshape(lead + (B, ))

    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        return self._eval(value, 0)

    def prob(self, value: torch.Tensor) -> torch.Tensor:
        return self._eval(value, 1)

    def log_prob_unnormalized(self, value: torch.Tensor) -> torch.Tensor:
        if self.beta is None:
            raise RuntimeError('log_prob_unnormalized is defined for the exponential family only')
        return self._eval(value, 2)

    def get_max_log_prob(self) -> torch.Tensor:
        grid = torch.tensor(fibonacci_grid(4096), dtype=torch.float32, device=self.device).unsqueeze(1)
        return self.log_prob(grid).max(dim=0).values

    def get_max_prob(self) -> torch.Tensor:
        grid = torch.tensor(fibonacci_grid(1024), dtype=torch.float32, device=self.device).unsqueeze(1)
        return self.prob(grid).max(dim=0).values

    @staticmethod
    def _uniform_sphere(shape, device) -> torch.Tensor:
        theta = torch.acos(1 - 2 * torch.rand(shape, device=device))
        phi = 2 * math.pi * torch.rand(shape, device=device)
        return torch.stack([theta.sin() * phi.cos(), theta.sin() * phi.sin(), theta.cos()], dim=-1)

    def sample(self, sample_shape=torch.Size()) -> torch.Tensor:
        """Rejection sampling against a uniform proposal with the grid envelope of the reference
        (spherical_dists.py:105-140, 226-260); candidates come from torch's device generator."""
        sample_shape = torch.Size(sample_shape)
        num = int(np.prod(sample_shape)) if len(sample_shape) else 1
        B = self._summed.shape[0]
        log_u = math.log(1 / (4 * math.pi))
        if self.beta is not None:
            log_m = self.get_max_log_prob() - log_u
            m_max = float(torch.exp(log_m.clamp(-8, 8)).max())
        else:
            log_m = torch.log(self.get_max_prob()) - log_u
            m_max = float(torch.exp(log_m).max())
        count = min(max(1, int(2 * m_max)), 1024)
        got = torch.zeros(B, dtype=torch.long, device=self.device)
        res = torch.zeros(num, B, 3, dtype=torch.float32, device=self.device)
        while bool((got < num).any()):
            cand = self._uniform_sphere((count, B), self.device)
            thr = torch.exp(self.log_prob(cand) - log_m - log_u)
            acc = torch.rand(count, 1, device=self.device) < thr
            for b in torch.nonzero(got < num).flatten().tolist():
                take = cand[acc[:, b], b][:num - int(got[b])]
                res[int(got[b]):int(got[b]) + len(take), b] = take
                got[b] += len(take)
        return res.reshape(sample_shape + (B, 3))

    def argmax(self, count: Optional[int] = None) -> torch.Tensor:
        """best of `count` accepted draws: 128 for the exponential family, 256 otherwise
        (spherical_dists.py:149, 262)."""
        count = count or (128 if self.beta is not None else 256)
        samples = self.sample(torch.Size((count, )))
        score = self.log_prob_unnormalized(samples) if self.beta is not None else self.prob(samples)
        best = torch.argmax(score, dim=0).view(1, -1, 1).expand(1, -1, 3)
        return torch.gather(samples, 0, best).squeeze(0)


class StepDists(Sequence):
    """[focus_dist, element_dist, distance_dist, so3_dist] of one step() call.  `block` is the packed copy of the
    head outputs (mg_cov_head_outputs, one small launch inside step()); the torch objects are built on first access."""

    def __init__(self, ac, cfg, block: torch.Tensor, bags: torch.Tensor):
        self._args = (cfg.B, cfg.N, cfg.Z, cfg.G, ac.min_distance, ac.max_distance, ac.beta)
        off, _ = ac.slot_table['distance_log_stds']
        self._log_stds = ac.theta.detach()[off:off + cfg.G].clone()
        self._block, self._bags = block, bags
        self._built: Optional[list] = None

    @staticmethod
    def block_floats(cfg) -> int:
        return cfg.B * (cfg.N + 1 + cfg.Z + 2 * cfg.G + 200 + 1)

    def _build(self) -> list:
        B, N, Z, G, dmin, dmax, beta = self._args
        blk = self._block
        o = 0
        logits = blk[o:o + B * N].view(B, N); o += B * N
        natoms = blk[o:o + B].long(); o += B
        element_logits = blk[o:o + B * Z].view(B, Z); o += B * Z
        dout = blk[o:o + B * 2 * G].view(B, 2 * G); o += B * 2 * G
        coef = blk[o:o + B * 200].view(B, 25, 4, 2); o += B * 200
        log_z = blk[o:o + B]
        real = torch.arange(N, device=blk.device).unsqueeze(0) < natoms.unsqueeze(1)   # atom_mask (B, N)
        focus_mask = real.clone()
        focus_mask[:, 0] = True                                                        # agent.py:184-189
        focus_dist = D.Categorical(probs=masked_softmax(logits, focus_mask))
        element_dist = D.Categorical(probs=masked_softmax(element_logits, self._bags > 0))
        half_w, center = (dmax - dmin) / 2, (dmax + dmin) / 2
        stds = torch.exp(self._log_stds).clamp(1e-6)
        distance_dist = GaussianMixtureModel(log_probs=dout[:, :G], means=torch.tanh(dout[:, G:]) * half_w + center,
                                             stds=stds.expand(B, G))
        cond = SO3VecLite(coef[:, l * l:(l + 1) * (l + 1)].permute(0, 2, 1, 3).contiguous() for l in range(5))
        so3_dist = SO3DensityHIP(cond, beta, log_z if beta is not None else None, natoms == 0)
        return [focus_dist, element_dist, distance_dist, so3_dist]

    def _get(self) -> list:
        if self._built is None:
            with torch.no_grad():
                self._built = self._build()
        return self._built

    def __len__(self) -> int:
        return 4

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())
