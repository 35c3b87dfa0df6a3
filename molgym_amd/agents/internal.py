"""Internal-coordinate (SchNet) actor-critic on the gfx950 HIP kernels.

Drop-in for /root/reference/molgym/agents/internal/agent.py:17-353: same constructor keywords
(model_util.py:18-24), same ``step(observations, actions)`` contract with the 7-column action rows
``[stop, focus, element, distance, angle, dihedral, kappa]`` (agent.py:26,306-308).  The reference runs the
schnetpack SchNet embedding 3*B times per call at batch size 1 (agent.py:128,177); here the host builds ONE
ragged batch of 3B molecules -- the canvases and the canvases plus the hypothetical new atom at +/- dihedral,
placed by the z-matrix helper (internal/zmat.py:66-133, float64 numpy, no gradient, as in the reference which
goes through ``to_numpy``) -- and the whole step is two C-ABI calls.  ``step(observations)`` (rollouts) draws the
sub-actions stage by stage from the same kernels (see ``_step_sample``).
"""
import ctypes as C
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib, layout
from ..spaces import ActionSpace, ObservationSpace, ObservationType
from .base import FlatThetaAgent
from .covariant import _ptr, _stream, compact_canvases, observation_arrays


def place_new_atoms(pos: np.ndarray, natoms: np.ndarray, focus: np.ndarray, distance, angle, dihedral) -> np.ndarray:
    """Vectorised zmat.position_atom_helper: pos (B, N, 3) float64 (real atoms first) -> new positions (B, 3)."""
    B, N, _ = pos.shape
    out = np.zeros((B, 3))
    if np.any((focus >= natoms) & (natoms > 0)) or np.any(focus < 0):
        raise RuntimeError('Focus greater than number of atoms')
    nz = np.nonzero(natoms > 0)[0]
    if len(nz) == 0:
        return out
    p, n, f = pos[nz], natoms[nz], focus[nz]
    fpos = p[np.arange(len(nz)), f]
    dist = np.sqrt(np.sum(np.square(p - fpos[:, None, :]), axis=-1))
    dist = np.where(np.arange(N)[None, :] < n[:, None], dist, np.inf)
    order = np.argsort(dist, axis=1, kind='stable')  # sorted() in the reference is stable too
    rows = np.arange(len(nz))
    p2 = p[rows, order[:, 0]]
    p1 = np.where((n >= 2)[:, None], p[rows, order[:, min(1, N - 1)]], p2 + np.array([1.0, 0.0, 0.0]))
    p0_three = p[rows, order[:, min(2, N - 1)]]
    p0 = np.where((n >= 3)[:, None], p0_three,
                  np.where((n == 2)[:, None], p2 + p1 + np.array([1.0, 1.0, 0.0]), p2 + np.array([0.0, 1.0, 0.0])))
    d, a, h = distance[nz, None], angle[nz, None], dihedral[nz, None]
    x, y, z = d * np.cos(a), d * np.cos(h) * np.sin(a), d * np.sin(h) * np.sin(a)
    v_a = p1 - p0
    v_b = p2 - p1
    v_b = v_b / np.linalg.norm(v_b, axis=1, keepdims=True)
    c_ab = np.cross(v_a, v_b)
    c_ab = c_ab / np.linalg.norm(c_ab, axis=1, keepdims=True)
    c_ab_b = np.cross(c_ab, v_b)
    out[nz] = p2 - v_b * x + c_ab_b * y + c_ab * z
    return out


class IntBatch:
    def __init__(self, cfg, mol_off, edge_off, molZ, molpos, bags, actions, logp=None, adv=None, ret=None):
        self.cfg, self.mol_off, self.edge_off, self.molZ, self.molpos = cfg, mol_off, edge_off, molZ, molpos
        self.bags, self.actions = bags, actions
        self.logp, self.adv, self.ret = logp, adv, ret


class IntRollout:
    """A rollout prepared for `ppo.train`'s device path.  The ragged 3B-molecule batch of a mini-batch depends on
    which samples are in it (z-matrix placement per sample, molecule / pair offsets), so it is assembled on the host
    per mini-batch; the float64 loss inputs are parked in HBM once and gathered on the device."""

    def __init__(self, ac, data):
        dev = ac.theta.device
        self.ac = ac
        # parsed ONCE for the whole rollout, together with the z-matrix placements of the recorded actions: a mini-batch is
        # then numpy slicing + the ragged assembly (the per-mini-batch parse + placement was 0.8 ms of host time)
        self.parsed = ac._parse(data['obs'])
        self.placed = ac._placements(self.parsed, np.asarray(data['act']))
        f64 = lambda x: x.to(dev) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, dtype=np.float64)).to(dev)
        self.logp, self.adv, self.ret = f64(data['logp']), f64(data['adv']), f64(data['ret'])

    def minibatch(self, indices: np.ndarray, idx_dev: Optional[torch.Tensor] = None) -> IntBatch:
        idx = np.asarray(indices, dtype=np.int64)
        batch = self.ac._assemble(tuple(x[idx] for x in self.parsed), tuple(x[idx] for x in self.placed))
        if idx_dev is None:
            idx_dev = torch.from_numpy(idx).to(self.logp.device)
        with torch.cuda.device(self.logp.device):
            batch.logp, batch.adv, batch.ret = _lib.gather_rows((self.logp, self.adv, self.ret), idx_dev,
                                                                _stream(self.logp.device))
        return batch


class _IntStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, ac, batch):
        lib = _lib.lib()
        nbytes = C.c_size_t()
        _lib.check(lib.mg_int_workspace_bytes(C.byref(batch.cfg), C.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=theta.device)
        out = torch.empty(3, batch.cfg.B, dtype=torch.float32, device=theta.device)
        with torch.cuda.device(theta.device):
            _lib.check(lib.mg_int_forward(C.byref(batch.cfg), _ptr(theta), _ptr(batch.mol_off), _ptr(batch.edge_off),
                                          _ptr(batch.molZ), _ptr(batch.molpos), _ptr(batch.bags), _ptr(batch.actions),
                                          _ptr(ws), nbytes.value, _ptr(out), _stream(theta.device)))
        ctx.save_for_backward(theta, ws)
        ctx.batch = batch
        ac._last_ws = ws
        return out

    @staticmethod
    def backward(ctx, gout):
        theta, ws = ctx.saved_tensors
        b = ctx.batch
        grad = torch.zeros_like(theta)
        gout = gout.contiguous()
        with torch.cuda.device(theta.device):
            _lib.check(_lib.lib().mg_int_backward(C.byref(b.cfg), _ptr(theta), _ptr(b.mol_off), _ptr(b.edge_off),
                                                  _ptr(b.molZ), _ptr(b.molpos), _ptr(b.bags), _ptr(b.actions),
                                                  _ptr(ws), ws.numel(), _ptr(gout), _ptr(grad),
                                                  _stream(theta.device)))
        return grad, None, None


class SchNetAC(FlatThetaAgent):
    def __init__(self, observation_space: ObservationSpace, action_space: ActionSpace,
                 min_max_distance: Tuple[float, float], network_width: int, device=None):
        super().__init__(observation_space, action_space)
        self.device = torch.device(device) if device is not None else torch.device('cuda')
        self.zs = list(self.observation_space.zs)
        self.num_atoms = self.observation_space.canvas_space.size
        self.num_zs = len(self.zs)
        self.network_width = network_width
        self.min_distance, self.max_distance = min_max_distance
        self.slot_table, total = layout.offsets_internal(self.num_zs, network_width)
        self.theta = torch.nn.Parameter(self._init_theta(total))
        self._last_ws = None
        self.to(self.device)

    def __getstate__(self):  # whole-module pickling (tools/model_util.py:82-91): drop the workspace cache
        state = self.__dict__.copy()
        state['_last_ws'] = None
        state.pop('_ws_cache', None)
        state.pop('_ws_epoch', None)
        return state

    def _init_theta(self, total: int) -> torch.Tensor:
        """schnetpack initialisers (Embedding N(0,1) with a zero padding row, Dense = xavier_uniform + zero
        bias), orthogonal MLPs with zero bias (modules.py:30-34), log stds of agent.py:69-70."""
        theta = torch.zeros(total)
        for name, (off, shape) in self.slot_table.items():
            n = int(np.prod(shape))
            view = theta[off:off + n].view(shape)
            if name == 'embedding_fn.embedding.weight':
                view.normal_()
                view[0].zero_()
            elif name.startswith('embedding_fn') and name.endswith('weight'):
                torch.nn.init.xavier_uniform_(view)
            elif name == 'log_stds':
                view.copy_(torch.log(torch.tensor([0.15, 0.25, 0.25])))
            elif name.endswith('weight'):
                torch.nn.init.orthogonal_(view)
            # every bias starts at zero
        return theta

    def _parse(self, observations: List[ObservationType]):
        """What make_batch needs from the observations alone (the rollout step parses once for its five passes)."""
        # exact float64 positions for the z-matrix step, as the reference (ase.Atoms positions are float64); vectorised, and
        # a `ParsedObservations` (arrays already) is taken as is
        labels, xyz, bags = observation_arrays(observations, self.zs, self.num_atoms)
        pos64, charges, natoms = compact_canvases(labels, xyz, self.zs)
        bags = bags.astype(np.float32)
        return charges, bags, natoms, pos64

    def _placements(self, parsed, actions: np.ndarray):
        """validated action rows (float32) and the two hypothetical placements (dihedral kept / flipped) they imply"""
        N = self.num_atoms
        charges, bags, natoms, pos64 = parsed
        acts = np.ascontiguousarray(np.asarray(actions, dtype=np.float32))
        assert acts.shape == (len(natoms), 7)
        focus, element = np.rint(acts[:, 1]).astype(np.int64), np.rint(acts[:, 2]).astype(np.int64)
        if focus.min() < 0 or focus.max() >= N or element.min() < 0 or element.max() >= self.num_zs:
            raise RuntimeError('index out of range in one-hot selection')
        a64 = acts.astype(np.float64)  # the agent casts actions to its dtype
        new_p = place_new_atoms(pos64, natoms, focus, a64[:, 3], a64[:, 4], a64[:, 5])
        new_m = place_new_atoms(pos64, natoms, focus, a64[:, 3], a64[:, 4], -a64[:, 5])
        z_new = np.asarray(self.zs, dtype=np.int32)[element]
        return acts, new_p, new_m, z_new

    def _assemble(self, parsed, placed) -> IntBatch:
        """the ragged 3B-molecule batch (canvas, canvas + new atom at +/- dihedral) of B parsed samples; ONE upload"""
        N = self.num_atoms
        charges, bags, natoms, pos64 = parsed
        acts, new_p, new_m, z_new = placed
        B = len(natoms)
        real = np.arange(N)[None, :] < natoms[:, None]
        base_z, base_p = charges[real], pos64[real]
        TA = int(natoms.sum())
        sizes = np.concatenate([natoms, natoms + 1, natoms + 1]).astype(np.int64)
        mol_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        edge_off = np.concatenate([[0], np.cumsum(sizes * (sizes - 1))]).astype(np.int32)
        MA = int(mol_off[-1])
        molZ, molpos = np.zeros(MA, dtype=np.int32), np.zeros((MA, 3))
        molZ[:TA], molpos[:TA] = base_z, base_p
        atom_b = np.repeat(np.arange(B), natoms)
        local = np.arange(TA) - np.repeat(np.cumsum(natoms) - natoms, natoms)
        for s, new in ((1, new_p), (2, new_m)):
            start = mol_off[s * B:(s + 1) * B].astype(np.int64)
            molZ[start[atom_b] + local], molpos[start[atom_b] + local] = base_z, base_p
            molZ[start + natoms], molpos[start + natoms] = z_new, new
        cfg = _lib.IntCfg()
        cfg.B, cfg.N, cfg.Z, cfg.W = B, N, self.num_zs, self.network_width
        for i, z in enumerate(self.zs):
            cfg.zs[i] = int(z)
        cfg.TA, cfg.MA, cfg.ME = TA, MA, int(edge_off[-1])
        cfg.min_distance, cfg.max_distance = float(self.min_distance), float(self.max_distance)
        # all six arrays are 4-byte typed: one packed host buffer, one copy, six views (each 16-byte aligned)
        parts = (mol_off, edge_off, molZ, molpos.astype(np.float32).reshape(-1), np.asarray(bags, dtype=np.float32).reshape(-1),
                 acts.reshape(-1))
        offs, total = [], 0
        for x in parts:
            offs.append(total)
            total += (x.size + 3) // 4 * 4
        host = np.zeros(max(total, 4), dtype=np.int32)
        for x, o in zip(parts, offs):
            host[o:o + x.size] = x.view(np.int32)
        devbuf = torch.from_numpy(host).to(self.theta.device)
        v = [devbuf[o:o + x.size] for x, o in zip(parts, offs)]
        return IntBatch(cfg, v[0], v[1], v[2], v[3].view(torch.float32).view(MA, 3), v[4].view(torch.float32).view(B, -1),
                        v[5].view(torch.float32).view(B, 7))

    def make_batch(self, observations: List[ObservationType], actions: np.ndarray, parsed=None) -> IntBatch:
        if parsed is None:
            parsed = self._parse(observations)
        return self._assemble(parsed, self._placements(parsed, actions))

    def to_action_space(self, action: np.ndarray, observation: ObservationType):
        """(atomic-number index, position) of the atom the 7-column action row places (agent.py:91-110)."""
        stop, focus, element, distance, angle, dihedral, kappa = (float(x) for x in action)
        if stop:
            return None  # the reference builds an empty ase.Atoms here; this agent never stops (agent.py:190-191)
        focus, element = int(round(focus)), int(round(element))
        sign = -1.0 if int(round(kappa)) else 1.0
        atoms = [xyz for label, xyz in observation[0] if self.zs[label] != 0]  # ObservationSpace.parse drops nulls
        n = len(atoms)
        pos = np.zeros((1, max(self.num_atoms, 1), 3))
        for k, xyz in enumerate(atoms):
            pos[0, k] = xyz
        new = place_new_atoms(pos, np.array([n]), np.array([focus]), np.array([distance]), np.array([angle]),
                              np.array([sign * dihedral]))[0]
        index = self.action_space.zs.index(self.observation_space.zs[element])
        return index, tuple(float(x) for x in new)

    def _actions_to_space(self, acts: np.ndarray, pos64: np.ndarray, natoms: np.ndarray, placed=None):
        """to_action_space for a whole batch of action rows: ONE vectorised z-matrix placement instead of one per sample
        (140 single placements are 13 ms of numpy call overhead; this is 0.2 ms) -- same arithmetic, same float64 inputs.
        `placed` = (new_plus, new_minus) of `_placements` for the same rows: the kept / flipped placement is picked, not redone."""
        a64 = np.asarray(acts, dtype=np.float32).astype(np.float64)
        focus, element = np.rint(a64[:, 1]).astype(np.int64), np.rint(a64[:, 2]).astype(np.int64)
        if placed is not None:
            new = np.where((np.rint(a64[:, 6]) != 0)[:, None], placed[1], placed[0])
        else:
            sign = np.where(np.rint(a64[:, 6]) != 0, -1.0, 1.0)
            new = place_new_atoms(pos64, natoms, focus, a64[:, 3], a64[:, 4], sign * a64[:, 5])
        out = []
        for b in range(len(a64)):
            if a64[b, 0]:
                out.append(None)
                continue
            index = self.action_space.zs.index(self.observation_space.zs[int(element[b])])
            out.append((index, tuple(float(x) for x in new[b])))
        return out

    def _forward_nograd(self, batch: IntBatch):
        lib = _lib.lib()
        nbytes = C.c_size_t()
        _lib.check(lib.mg_int_workspace_bytes(C.byref(batch.cfg), C.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=self.theta.device)
        out = torch.empty(3, batch.cfg.B, dtype=torch.float32, device=self.theta.device)
        with self._guard():
            _lib.check(lib.mg_int_forward(C.byref(batch.cfg), _ptr(self.theta), _ptr(batch.mol_off),
                                          _ptr(batch.edge_off), _ptr(batch.molZ), _ptr(batch.molpos), _ptr(batch.bags),
                                          _ptr(batch.actions), _ptr(ws), nbytes.value, _ptr(out), self._s()))
        self._last_ws = ws
        return out, ws

    def prepare_rollout(self, data: Dict[str, Any]) -> IntRollout:
        return IntRollout(self, data)

    def prepare_batch(self, observations, actions, logp=None, adv=None, ret=None) -> IntBatch:
        batch = self.make_batch(observations, actions)
        dev = self.theta.device
        f64 = lambda x: None if x is None else torch.as_tensor(np.asarray(x, dtype=np.float64)).to(dev)
        batch.logp, batch.adv, batch.ret = f64(logp), f64(adv), f64(ret)
        return batch

    # ---- what depends on theta alone, once per EPOCH (ppo.py:117-146: one optimizer step per epoch), as CovariantAC does ----
    def invalidate_weights(self) -> None:
        """theta may have changed (start of a PPO epoch): the derived weight matrices cached in the workspaces are stale"""
        for st in self.__dict__.get('_ws_epoch', {}).values():
            st['weights'] = False

    def fold_gradients(self) -> None:
        """(interface of ppo.train's epoch loop: this agent has no expanded weight gradients to fold -- theta.grad is complete
        after every mini-batch)"""

    def _step_workspace(self, cfg, slot: int) -> torch.Tensor:
        """one cached block per slot (the derived weights sit first in it, at offsets that do not depend on the batch)"""
        nbytes = C.c_size_t()
        _lib.check(_lib.lib().mg_int_workspace_bytes(C.byref(cfg), C.byref(nbytes)))
        cache = self.__dict__.setdefault('_ws_cache', {})
        ws = cache.get(slot)
        if ws is None or ws.numel() < nbytes.value or ws.device != self.theta.device:
            ws = torch.empty(int(nbytes.value * 1.25), dtype=torch.uint8, device=self.theta.device)
            cache[slot] = ws
            self.__dict__.setdefault('_ws_epoch', {}).pop(slot, None)
        ws.record_stream(torch.cuda.current_stream(self.theta.device))
        return ws

    def ppo_minibatch(self, batch: IntBatch, clip_ratio: float, vf_coef: float, entropy_coef: float,
                      loss_scale: float = 1.0, slot: int = 0, stats_accum: Optional[torch.Tensor] = None,
                      graph: Optional[bool] = None, epoch_cache: bool = False) -> torch.Tensor:
        """forward + float64 PPO loss + hand-written backward on the device (ppo.py:124-131) in ONE C call (mg_int_ppo_step),
        gradients accumulated into theta.grad (scaled by `loss_scale`: the data-parallel B_local / B_global); same signature
        and meaning as CovariantAC.ppo_minibatch: `stats_accum` receives loss_scale x statistics on the device, `graph` (default
        on) issues the launches as one hipGraph launch whose kernel nodes are updated in place, one cached graph per `slot`;
        `epoch_cache` (ppo.train's loop): the derived weight matrices of this slot's workspace are prepared by the slot's FIRST
        mini-batch after `invalidate_weights()` only.  Returns the 6 loss statistics (float64 device tensor, no sync)."""
        lib = _lib.lib()
        B = batch.cfg.B
        dev = self.theta.device
        ws = self._step_workspace(batch.cfg, slot)
        flags = 0
        if epoch_cache:
            st = self.__dict__.setdefault('_ws_epoch', {}).setdefault(slot, {'weights': False})
            flags = _lib.STEP_WEIGHTS_CURRENT if st['weights'] else 0
            st['weights'] = True
        else:
            self.__dict__.get('_ws_epoch', {}).pop(slot, None)
        out = torch.empty(3, B, dtype=torch.float32, device=dev)
        stats = torch.empty(6, dtype=torch.float64, device=dev)
        gout = torch.empty(3, B, dtype=torch.float32, device=dev)
        if self.theta.grad is None:
            self.theta.grad = torch.zeros_like(self.theta)
        use_graph = getattr(self, 'use_graphs', True) if graph is None else graph
        used = C.c_int32(0)
        with self._guard():
            _lib.check(lib.mg_int_ppo_step(C.byref(batch.cfg), _ptr(self.theta), _ptr(batch.mol_off), _ptr(batch.edge_off),
                                           _ptr(batch.molZ), _ptr(batch.molpos), _ptr(batch.bags), _ptr(batch.actions), _ptr(ws),
                                           ws.numel(), _ptr(batch.logp), _ptr(batch.adv), _ptr(batch.ret), clip_ratio, vf_coef,
                                           entropy_coef, float(loss_scale), _ptr(out), _ptr(gout), _ptr(stats), _ptr(stats_accum),
                                           _ptr(self.theta.grad), slot if use_graph else -1, flags, C.byref(used), self._s()))
        self._last_ws = ws
        self.last_step_used_graph = bool(used.value)
        return stats

    def _ws_view(self, cfg, ws: torch.Tensor, name: str) -> torch.Tensor:
        off, cnt = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().mg_int_workspace_lookup(C.byref(cfg), name.encode(), C.byref(off), C.byref(cnt)))
        return ws.view(torch.float32)[off.value:off.value + cnt.value]

    def _step_sample(self, observations: List[ObservationType]) -> Dict[str, Any]:
        """step(obs) of the rollout (agent.py:181-353 with actions=None).  The sub-actions are conditioned on each
        other (element on the focus, the continuous means on both, kappa on the placed atom), so the evaluation
        kernels run once per stage on the partially filled action rows; every draw uses torch's device RNG
        (Categorical / Normal as in the reference, so torch.manual_seed governs rollouts), argmax / means in
        evaluation mode.  The last pass is the plain evaluation of the completed rows, so logp / ent / v are
        exactly what step(obs, actions) returns for them."""
        B, N, Z = len(observations), self.num_atoms, self.num_zs
        dev = self.theta.device
        parsed = self._parse(observations)
        _, bags, natoms, pos64 = parsed
        acts = np.zeros((B, 7), dtype=np.float32)  # stop = 0: this agent does not stop
        acts[:, 3] = 0.5 * (self.min_distance + self.max_distance)  # harmless placeholders for the early passes
        acts[:, 4] = acts[:, 5] = 0.5 * np.pi
        nat = torch.from_numpy(natoms.astype(np.int64)).to(dev)
        with torch.no_grad():
            # [r5] focus, element and the three continuous sub-actions only read the BASE molecules' latents (the +/- dihedral
            # copies feed the kappa head alone), so their three passes share ONE assembled batch whose action rows are completed on
            # the device as the draws are made; the placements are computed once all of them are known: two assemblies and uploads
            # and one device -> host copy where there were five and four
            # focus (agent.py:206-221): softmax over the real atoms; an empty canvas focuses slot 0
            # (the first assembly's +/- copies get their new atom far outside every cutoff: nothing reads their latents yet)
            far = 1.0e3 + 10.0 * np.arange(B, dtype=np.float64)[:, None] * np.ones((1, 3))
            batch = self._assemble(parsed, (acts, far, far, np.full(B, self.zs[0], dtype=np.int32)))
            _, ws = self._forward_nograd(batch)
            logit_f = self._ws_view(batch.cfg, ws, 'logitF')[:batch.cfg.TA]
            dense = torch.full((B, N), float('-inf'), device=dev)
            slot = torch.arange(N, device=dev)[None, :] < nat[:, None]
            dense[slot] = logit_f
            dense[nat == 0, 0] = 0.0
            p_f = torch.softmax(dense, dim=-1)
            # (torch.multinomial / torch.normal directly: what Categorical.sample / Normal.sample call, same draws from the same RNG stream)
            focus = torch.multinomial(p_f / p_f.sum(-1, keepdim=True), 1, True).squeeze(1) if self.training else torch.argmax(p_f, -1)
            batch.actions[:, 1] = focus.to(torch.float32)
            # element (agent.py:229-242): softmax over the elements left in the bag
            _, ws = self._forward_nograd(batch)
            logit_e = self._ws_view(batch.cfg, ws, 'logitE')[:B * Z].view(B, Z)
            mask_e = torch.from_numpy(bags > 0).to(dev)
            dense = torch.where(mask_e, logit_e, torch.full_like(logit_e, float('-inf')))
            dense[~mask_e.any(dim=-1), 0] = 0.0  # an exhausted bag never reaches the agent; keep the draw defined
            p_e = torch.softmax(dense, dim=-1)
            element = torch.multinomial(p_e / p_e.sum(-1, keepdim=True), 1, True).squeeze(1) if self.training else torch.argmax(p_e, -1)
            batch.actions[:, 2] = element.to(torch.float32)
            # distance / angle / dihedral (agent.py:246-292): Normal around tanh(mean) * width / 2 + center
            _, ws = self._forward_nograd(batch)
            cout = self._ws_view(batch.cfg, ws, 'cout')[:B * 3].view(B, 3)
            half_w = torch.tensor([0.5 * (self.max_distance - self.min_distance), 0.5 * np.pi, 0.5 * np.pi], device=dev)
            center = torch.tensor([0.5 * (self.max_distance + self.min_distance), 0.5 * np.pi, 0.5 * np.pi], device=dev)
            mean = torch.tanh(cout) * half_w + center
            if self.training:
                o, shp = self.slot_table['log_stds']
                scale = torch.exp(1e-6 + self.theta[o:o + 3])
                cont = torch.normal(mean, scale.expand_as(mean))
                cont[:, 0].clamp_(min=0.001)  # the sampled distance must stay positive (agent.py:254-255)
            else:
                cont = mean
            drawn = torch.cat([focus.to(torch.float32)[:, None], element.to(torch.float32)[:, None], cont.to(torch.float32)], dim=1)
            acts[:, 1:6] = drawn.cpu().numpy()
            # kappa (agent.py:294-315): keep / flip the dihedral, logits from the two hypothetical placements
            placed = self._placements(parsed, acts)
            batch = self._assemble(parsed, placed)
            _, ws = self._forward_nograd(batch)
            kv = self._ws_view(batch.cfg, ws, 'kv')[:2 * B].view(2, B).t()
            kappa = torch.multinomial(torch.softmax(kv, dim=-1), 1, True).squeeze(1) if self.training else torch.argmax(kv, -1)
            batch.actions[:, 6] = kappa.to(torch.float32)
            out, _ = self._forward_nograd(batch)  # the plain evaluation of the completed rows
            acts[:, 6] = kappa.cpu().numpy()
        return {'a': batch.actions, 'logp': out[0], 'ent': out[1], 'v': out[2],
                'actions': self._actions_to_space(acts, pos64, natoms, placed[1:3])}

    def step(self, observations: List[ObservationType], actions: Optional[np.ndarray] = None) -> Dict[str, Any]:
        if self.theta.device.type != 'cuda':
            raise RuntimeError('SchNetAC runs on the HIP device only (no CPU fallback)')
        if actions is None:
            return self._step_sample(observations)
        parsed = self._parse(observations)
        batch = self.make_batch(observations, actions, parsed)
        out = _IntStep.apply(self.theta, self, batch)
        acts = np.asarray(actions, dtype=np.float32)
        return {'a': batch.actions, 'logp': out[0], 'ent': out[1], 'v': out[2],
                'actions': self._actions_to_space(acts, parsed[3], parsed[2])}
