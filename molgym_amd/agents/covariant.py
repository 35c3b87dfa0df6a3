"""Covariant (Cormorant) actor-critic on the gfx950 HIP kernels.

Drop-in for /root/reference/molgym/agents/covariant/agent.py:20-334: same
constructor keywords (model_util.py:26-39), same ``step(observations, actions)``
contract (base.py:17-19).  All arithmetic of ``step`` with actions given -- the PPO
training path, ppo.py:26 -- runs in libmolgym_hip.so through one autograd node;
there is no eager / CPU fallback.
"""
import ctypes as C
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib, layout
from ..lebedev import lebedev_table
from ..observations import ParsedObservations
from ..spaces import ActionSpace, ObservationSpace, ObservationType
from .base import FlatThetaAgent
from .dists import StepDists


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream(device=None):
    """torch's current stream ON THE AGENT'S DEVICE (not on whatever device happens to be current)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def observation_arrays(observations, zs: List[int], canvas_size: int):
    """(labels (B, N) int64, xyz (B, N, 3) float64, bags (B, Z) int64) of a list of observation tuples or of a
    `ParsedObservations` (already arrays: no Python-object traversal), with the reference's shape / range errors."""
    po = ParsedObservations.from_list(observations, canvas_size, len(zs))
    B = len(po)
    if po.labels.shape != (B, canvas_size) or po.xyz.shape != (B, canvas_size, 3):
        raise RuntimeError(f'canvas shape {po.labels.shape} does not match canvas_size {canvas_size}')
    if po.bags.shape != (B, len(zs)):
        raise RuntimeError(f'bag shape {po.bags.shape} does not match len(zs) {len(zs)}')
    if po.labels.min(initial=0) < 0 or po.labels.max(initial=0) >= len(zs):
        raise RuntimeError('Invalid atomic number index in canvas')
    return po.labels, po.xyz, po.bags


def compact_canvases(labels: np.ndarray, xyz: np.ndarray, zs: List[int]):
    """null items dropped, real atoms compacted to the front in canvas order, zero padded (covariant/tools.py:8-49):
    (xyz float64 (B, N, 3), charges int32 (B, N), natoms (B,))"""
    charges = np.asarray(zs, dtype=np.int32)[labels]
    real = charges > 0
    order = np.argsort(~real, axis=1, kind='stable')
    charges = np.take_along_axis(charges, order, axis=1)
    xyz = np.take_along_axis(xyz, order[..., None], axis=1)
    xyz[~np.take_along_axis(real, order, axis=1)] = 0.0
    return xyz, np.ascontiguousarray(charges), real.sum(axis=1)


def parse_observations_host(observations, zs: List[int], canvas_size: int):
    """Observation tuples (or a `ParsedObservations`) -> padded numpy arrays (agent.py:165-197, covariant/tools.py:8-49
    without ase): null items dropped, real atoms compacted to the front in canvas order, zero padded."""
    labels, xyz, bags = observation_arrays(observations, zs, canvas_size)
    xyz, charges, natoms = compact_canvases(labels, xyz, zs)
    return xyz.astype(np.float32), charges, bags.astype(np.float32), natoms


class DeviceBatch:
    """One PPO mini-batch resident in HBM (layout of include/molgym_hip.h)."""

    def __init__(self, cfg, pos, charges, bags, actions, logp=None, adv=None, ret=None):
        self.cfg, self.pos, self.charges, self.bags, self.actions = cfg, pos, charges, bags, actions
        self.logp, self.adv, self.ret = logp, adv, ret


class RolloutOnDevice:
    """A parsed rollout resident in HBM; `minibatch(indices)` gathers one PPO mini-batch on the device."""

    def __init__(self, ac, natoms, pos, charges, bags, actions, logp, adv, ret):
        self.ac, self.natoms = ac, natoms
        self.t = (pos, charges, bags, actions, logp, adv, ret)

    def minibatch(self, indices: np.ndarray, idx_dev: Optional[torch.Tensor] = None) -> DeviceBatch:
        """`idx_dev`: the same indices already on the device (ppo.train uploads an epoch's permutation once)"""
        if idx_dev is None:
            idx_dev = torch.as_tensor(np.asarray(indices, dtype=np.int64)).to(self.t[0].device)
        with torch.cuda.device(self.t[0].device):
            pos, charges, bags, actions, logp, adv, ret = _lib.gather_rows(self.t, idx_dev, _stream(self.t[0].device))
        return DeviceBatch(self.ac._make_cfg(len(indices), self.natoms[indices]), pos, charges, bags, actions, logp,
                           adv, ret)


class _CovStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, ac, cfg, pos, charges, bags, actions):
        lib = ac._L()
        nbytes = C.c_size_t()
        _lib.check(lib.mg_cov_workspace_bytes(C.byref(cfg), C.byref(nbytes)), lib)
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=theta.device)
        out = torch.empty(3, cfg.B, dtype=torch.float32, device=theta.device)
        with torch.cuda.device(theta.device):
            _lib.check(lib.mg_cov_forward(C.byref(cfg), _ptr(theta), _ptr(pos), _ptr(charges), _ptr(bags),
                                          _ptr(actions), _ptr(ac.leb), _ptr(ws), nbytes.value, _ptr(out),
                                          _stream(theta.device)), lib)
        ctx.save_for_backward(theta, pos, charges, bags, actions, ws)
        ctx.cfg, ctx.ac = cfg, ac
        ac._last_ws = ws
        return out

    @staticmethod
    def backward(ctx, gout):
        theta, pos, charges, bags, actions, ws = ctx.saved_tensors
        lib = ctx.ac._L()
        grad = torch.zeros_like(theta)
        gout = gout.contiguous()
        with torch.cuda.device(theta.device):
            _lib.check(lib.mg_cov_backward(C.byref(ctx.cfg), _ptr(theta), _ptr(pos), _ptr(charges), _ptr(bags),
                                           _ptr(actions), _ptr(ctx.ac.leb), _ptr(ws), ws.numel(), _ptr(gout),
                                           _ptr(grad), _stream(theta.device)), lib)
        return grad, None, None, None, None, None, None


class CovariantAC(FlatThetaAgent):
    def __init__(
        self,
        observation_space: ObservationSpace,
        action_space: ActionSpace,
        min_max_distance: Tuple[float, float],
        network_width: int,
        maxl: int,
        num_cg_levels: int,
        num_channels_hidden: int,
        num_channels_per_element: int,
        num_gaussians: int,
        bag_scale: int,
        beta: Optional[float] = None,
        device=None,
    ):
        super().__init__(observation_space, action_space)
        if not 1 <= int(maxl) <= layout.MAXL:
            raise RuntimeError(f'maxl {maxl}: the gfx950 kernels cover 1..{layout.MAXL} (the Clebsch-Gordan tables, thread maps and LDS '
                               f'layouts are laid out for the 25 (l, m) rows of maxl = 4, the reference default, arg_parser.py:56; a '
                               f'smaller maxl runs as the same network with the higher degrees structurally zero)')
        if not _lib.LEVELS_RANGE[0] <= int(num_cg_levels) <= _lib.LEVELS_RANGE[1]:
            raise RuntimeError(f'num_cg_levels {num_cg_levels}: the kernels cover {_lib.LEVELS_RANGE[0]}..{_lib.LEVELS_RANGE[1]}')
        # the channel counts and num_cg_levels are compile-time constants of a library build: the defaults load
        # libmolgym_hip.so, other values their own build of the same sources (compiled on first use,
        # molgym_amd/_lib.py::build_variant)
        self._channels = (int(num_channels_hidden), int(num_channels_per_element), int(num_cg_levels))
        self.device = torch.device(device) if device is not None else torch.device('cuda')
        self.dtype = torch.float
        self.zs = list(self.observation_space.zs)
        self.min_distance, self.max_distance = min_max_distance
        assert self.min_distance < self.max_distance
        self.beta = beta
        self.max_sh, self.num_cg_levels = maxl, num_cg_levels
        self.num_channels_hidden, self.num_channels_per_element = num_channels_hidden, num_channels_per_element
        if network_width > 1024 or network_width % 4:
            # (<= 128: all heads in one launch per direction; wider: the staged head kernels + row GEMMs, any multiple of 4)
            raise RuntimeError(f'network_width {network_width}: the HIP heads kernels support multiples of 4 up to 1024')
        self.num_gaussians, self.network_width, self.bag_scale = num_gaussians, network_width, bag_scale
        self.num_channels_out = len(self.zs) * num_channels_per_element
        self.slot_table, total = layout.offsets(len(self.zs), network_width, num_gaussians, *self._channels, int(maxl))  # (C, Ce, levels, maxl)
        self._L()  # load (build) the library for these channel counts now: a missing toolchain fails here, not mid-rollout
        self.theta = torch.nn.Parameter(self._init_theta(total))
        self.register_buffer('leb', torch.from_numpy(lebedev_table()), persistent=False)
        # [r6] maxl < 4 (arg_parser.py:56): theta / state_dict are the maxl-limited module's (the reference's shapes); the kernels
        # read the maxl = 4 layout, into which theta is EMBEDDED -- the smaller network is the larger one with every quantity of a
        # degree above maxl multiplied by a zero weight, so outputs are those of the smaller network and its gradient is the gather
        # of the larger one's (layout.embedding_index; exact: tests/test_host.py checks it on the oracle, the GPU tests against the
        # oracle built with the same maxl).  Not faster than maxl = 4: the kernels still walk the 25 (l, m) rows.
        if int(maxl) < layout.MAXL:
            _, total_k = layout.offsets(len(self.zs), network_width, num_gaussians, *self._channels)
            idx = layout.embedding_index(len(self.zs), network_width, num_gaussians, *self._channels, int(maxl))
            self.register_buffer('_embed_idx', torch.from_numpy(idx), persistent=False)
            self.register_buffer('_theta_k', torch.zeros(total_k), persistent=False)
        self._last_ws = None
        # ppo_minibatch issues its launches as one updated hipGraph launch (include/molgym_hip.h: mg_cov_ppo_step)
        self.use_graphs = True
        self.to(self.device)

    def _chk(self, rc):
        _lib.check(rc, self._L())

    def _L(self):
        """the library build for this agent's channel counts"""
        return _lib.lib(getattr(self, '_channels', None))

    # -- the parameter vector the KERNELS read (maxl = 4 layout) and the gradient buffer they add into ----------------------
    def _ktheta(self, refresh: bool = True) -> torch.Tensor:
        if self.max_sh == layout.MAXL:
            return self.theta
        if refresh:  # (positions outside the index are zero from construction and never written; rewriting equal values is
            with torch.no_grad():  # harmless to a kernel of another stream that reads them meanwhile)
                self._theta_k.index_copy_(0, self._embed_idx, self.theta.detach())
        return self._theta_k

    def _kgrad(self, slot: int = 0) -> torch.Tensor:
        if self.max_sh == layout.MAXL:
            return self.theta.grad
        bufs = self.__dict__.setdefault('_grad_k', {})
        if slot not in bufs or bufs[slot].device != self.theta.device:
            bufs[slot] = torch.zeros_like(self._theta_k)
        return bufs[slot]

    def _kgrad_collect(self, slot: int) -> None:
        """maxl < 4: theta.grad += gather of the slot's kernel-side gradient buffer (float atomics: mini-batches of other
        streams may be adding to theta.grad too), buffer zeroed for its next use"""
        if self.max_sh == layout.MAXL:
            return
        gk = self._grad_k[slot]
        iota = self.__dict__.get('_iota')
        if iota is None or iota.device != self.theta.device:
            iota = self._iota = torch.arange(self.theta.numel(), device=self.theta.device)
        self.theta.grad.index_add_(0, iota, gk.index_select(0, self._embed_idx))
        gk.zero_()

    # whole-module pickling (ModelIO.save = torch.save(module), tools/model_util.py:82-91): drop the caches
    def __getstate__(self):
        state = self.__dict__.copy()
        state['_last_ws'] = None
        state.pop('_last_cfg', None)
        state.pop('_ws_cache', None)
        state.pop('_ws_epoch', None)
        state.pop('_unchecked', None)
        state.pop('_last_out', None)
        state.pop('_grad_k', None)
        state.pop('_iota', None)
        return state

    # -- parameters -----------------------------------------------------------------------------
    def _init_theta(self, total: int) -> torch.Tensor:
        """Same initialisers as the reference stack: torch Linear defaults for the radial / input
        Linear layers, U(-1,1)*gain/max(dims) for the complex mixes ('rand'), orthogonal + zero bias
        for the MLPs (modules.py:30-34), log 0.1 for the GMM widths (agent.py:134-136)."""
        theta = torch.zeros(total)
        for name, (off, shape) in self.slot_table.items():
            n = int(np.prod(shape))
            view = theta[off:off + n].view(shape)
            if name.endswith('scales'):
                view.copy_(torch.tensor([0., 1., 2., 3., 0., 1., 2., 3.]).view(shape))
            elif name.endswith('phases'):
                view.copy_(torch.tensor([0.] * 4 + [np.pi / 2] * 4).view(shape))
            elif 'cat_mix.weights' in name:
                gain = 1.0 if 'edge_levels' in name else 10.0
                view.copy_((2 * torch.rand(shape) - 1) * (gain / max(shape)))
            elif name.startswith('phi_') and name.endswith('weight'):
                torch.nn.init.orthogonal_(view)
            elif name.startswith('phi_') and name.endswith('bias'):
                view.zero_()
            elif name == 'distance_log_stds':
                view.fill_(float(np.log(0.1)))
            elif ('.linear.' in name or name.endswith('lin.weight')) and name.endswith('weight'):
                lin = torch.nn.Linear(shape[1], shape[0])
                view.copy_(lin.weight.data)
                self._pending_bias = lin.bias.data.clone()
            elif name.endswith('bias'):
                view.copy_(self._pending_bias)
            else:
                raise RuntimeError(f'no initialiser for {name}')
        if hasattr(self, '_pending_bias'):
            del self._pending_bias
        return theta

    # -- batch ----------------------------------------------------------------------------------
    def _make_cfg(self, B: int, natoms: np.ndarray) -> _lib.CovCfg:
        cfg = _lib.CovCfg()
        cfg.B, cfg.N, cfg.Z = B, self.observation_space.canvas_space.size, len(self.zs)
        for i, z in enumerate(self.zs):
            cfg.zs[i] = int(z)
        cfg.W, cfg.G = self.network_width, self.num_gaussians
        cfg.TA, cfg.TE = int(natoms.sum()), int((natoms.astype(np.int64)**2).sum())
        cfg.has_beta = 0 if self.beta is None else 1
        cfg.beta = 0.0 if self.beta is None else float(self.beta)
        cfg.bag_scale = float(self.bag_scale)
        cfg.min_distance, cfg.max_distance = float(self.min_distance), float(self.max_distance)
        return cfg

    def to_action_space(self, action: np.ndarray, observation: ObservationType):
        """agent.py:147-163 without the ase round trip: position = focused atom + distance * direction, where the
        focus indexes the non-null canvas items in canvas order (ObservationSpace.parse drops the null items,
        spaces.py:55-61,106-107)."""
        assert action.shape == (6, )
        focus, element_index = int(round(float(action[0]))), int(round(float(action[1])))
        atoms = [xyz for label, xyz in observation[0] if self.zs[label] != 0]
        if len(atoms):
            position = tuple(np.asarray(atoms[focus], dtype=np.float64) + action[2] * action[3:6])
        else:
            position = (0.0, 0.0, 0.0)
        return element_index, position

    def _check_actions(self, actions, B: int) -> np.ndarray:
        acts = np.ascontiguousarray(np.asarray(actions, dtype=np.float32))
        assert acts.shape == (B, 6)
        N = self.observation_space.canvas_space.size
        focus, element = np.rint(acts[:, 0]), np.rint(acts[:, 1])
        if B and (focus.min() < 0 or focus.max() >= N or element.min() < 0 or element.max() >= len(self.zs)):
            raise RuntimeError('index out of range in one-hot selection')  # to_one_hot's scatter_ error
        return acts

    def _dists(self, cfg, ws: torch.Tensor, bags: torch.Tensor) -> StepDists:
        block = torch.empty(StepDists.block_floats(cfg), dtype=torch.float32, device=self.theta.device)
        with self._guard():
            self._chk(self._L().mg_cov_head_outputs(C.byref(cfg), _ptr(ws), ws.numel(), _ptr(block), self._s()))
        return StepDists(self, cfg, block, bags)

    def step(self, observations: List[ObservationType], actions: Optional[np.ndarray] = None) -> Dict[str, Any]:
        if self.theta.device.type != 'cuda':
            raise RuntimeError('CovariantAC runs on the HIP device only (no CPU fallback)')
        if actions is None:
            return self._step_sample(observations)
        N = self.observation_space.canvas_space.size
        pos, charges, bags, natoms = parse_observations_host(observations, self.zs, N)
        B = len(observations)
        acts = self._check_actions(actions, B)
        cfg = self._make_cfg(B, natoms)
        d_pos, d_chg, d_bag, d_act = self._upload_packed(pos, charges, bags, acts)
        theta = self.theta
        if self.max_sh != layout.MAXL:  # (differentiable embedding: autograd gathers the gradient back)
            theta = torch.zeros_like(self._theta_k).index_copy(0, self._embed_idx, self.theta)
        out = _CovStep.apply(theta, self, cfg, d_pos, d_chg, d_bag, d_act)
        return {'a': d_act, 'logp': out[0], 'ent': out[1], 'v': out[2],
                'dists': self._dists(cfg, self._last_ws, d_bag)}

    # -- device-resident mini-batches (the PPO fast path: no autograd graph, no host sync) ---------
    def prepare_batch(self, observations: List[ObservationType], actions: np.ndarray, logp=None, adv=None,
                      ret=None) -> 'DeviceBatch':
        """Parse on the host once and park everything a PPO mini-batch needs in HBM."""
        N = self.observation_space.canvas_space.size
        pos, charges, bags, natoms = parse_observations_host(observations, self.zs, N)
        B = len(observations)
        acts = self._check_actions(actions, B)
        dev = self.theta.device
        f64 = lambda x: None if x is None else torch.as_tensor(np.asarray(x, dtype=np.float64)).to(dev)
        d_pos, d_chg, d_bag, d_act = self._upload_packed(pos, charges, bags, acts)
        return DeviceBatch(self._make_cfg(B, natoms), d_pos, d_chg, d_bag, d_act, f64(logp), f64(adv), f64(ret))

    def prepare_rollout(self, data: Dict[str, Any]) -> 'RolloutOnDevice':
        """Parse a whole rollout (the `data` dict of ppo.train) once; mini-batches are device gathers."""
        N = self.observation_space.canvas_space.size
        pos, charges, bags, natoms = parse_observations_host(data['obs'], self.zs, N)
        acts = self._check_actions(data['act'], len(data['obs']))
        dev = self.theta.device
        f64 = lambda x: x.to(dev) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, dtype=np.float64)).to(dev)
        d_pos, d_chg, d_bag, d_act = self._upload_packed(pos, charges, bags, acts)
        return RolloutOnDevice(self, natoms, d_pos, d_chg, d_bag, d_act, f64(data['logp']), f64(data['adv']), f64(data['ret']))

    def _workspace(self, cfg: _lib.CovCfg, slot: int = 0, epoch_step: bool = False) -> torch.Tensor:
        nbytes = C.c_size_t()
        self._chk(self._L().mg_cov_workspace_bytes(C.byref(cfg), C.byref(nbytes)))
        cache = self.__dict__.setdefault('_ws_cache', {})
        ws = cache.get(slot)
        # Only THIS slot is folded here, on the current stream = the stream this slot's mini-batches were issued on
        # (ppo._DeviceRunner.run): the other slots' backward launches may still be adding to their accumulators on their own
        # streams, and k_fold_weights reads-then-zeroes non-atomically -- folding them from here would lose what lands in between.
        if not epoch_step and self.__dict__.get('_ws_epoch', {}).get(slot, {}).get('pending'):
            self.fold_gradients(slot)  # another kind of call on a workspace that still holds an epoch's deferred weight gradients
        if ws is None or ws.numel() < nbytes.value or ws.device != self.theta.device:
            if self.__dict__.get('_ws_epoch', {}).get(slot, {}).get('pending'):
                self.fold_gradients(slot)  # (the block is about to be replaced)
            ws = torch.empty(int(nbytes.value * 1.25), dtype=torch.uint8, device=self.theta.device)
            cache[slot] = ws
            self.__dict__.setdefault('_ws_epoch', {}).pop(slot, None)
        # the cached block may have been allocated under another stream (rollout: default stream; ppo.train: its own): tell
        # the caching allocator about THIS use, so that a later replacement is not handed out while kernels still read it
        ws.record_stream(torch.cuda.current_stream(self.theta.device))
        return ws

    # ---- what depends on theta alone, once per EPOCH instead of once per mini-batch (ppo.py:117-146: one optimizer step per epoch) ----
    def invalidate_weights(self) -> None:
        """theta may have changed (start of a PPO epoch): the derived weight matrices cached in the workspaces are stale"""
        self._ktheta()  # (maxl < 4: re-embed theta, on the caller's stream -- ppo._DeviceRunner.begin_epoch orders the others behind it)
        for st in self.__dict__.get('_ws_epoch', {}).values():
            st['weights'] = False
            # A slot still pending HERE was left by an epoch that never reached its fold (an exception between its mini-batches
            # and end_epoch): what its accumulator holds belongs to a gradient that was abandoned.  Drop the claim -- the slot's
            # next epoch_cache step prepares the weights anew, which zeroes the accumulator; a slot the new epoch never touches is
            # then NOT folded into the new theta.grad by end_epoch.
            st['pending'] = False

    def fold_gradients(self, slot: Optional[int] = None) -> None:
        """mg_cov_fold_grads for every workspace (`slot=None`) -- or the one of `slot` -- whose mini-batches ran with
        `epoch_cache=True` since the last fold: theta.grad is complete only after the all-slot call (ppo.train: once per epoch,
        before the all-reduce / norm / clip / Adam step).  Issued on the current stream, which must already be ordered behind
        the streams of the slots it folds (all of them: ppo._DeviceRunner.end_epoch after its wait_stream joins; one: the
        slot's own stream, _workspace)."""
        for sl, st in self.__dict__.get('_ws_epoch', {}).items():
            if slot is not None and sl != slot:
                continue
            slot_ = sl
            if st.get('pending'):
                ws = self._ws_cache[slot_]
                with self._guard():
                    self._chk(self._L().mg_cov_fold_grads(C.byref(st['cfg']), _ptr(ws), ws.numel(), _ptr(self._kgrad(slot_)), self._s()))
                self._kgrad_collect(slot_)
                ws.record_stream(torch.cuda.current_stream(self.theta.device))
                st['pending'] = False

    def forward_batch(self, batch: 'DeviceBatch', slot: int = 0) -> torch.Tensor:
        """(3, B) float32: logp, ent, v -- no autograd graph."""
        ws = self._workspace(batch.cfg, slot)
        out = torch.empty(3, batch.cfg.B, dtype=torch.float32, device=self.theta.device)
        with self._guard():
            self._chk(self._L().mg_cov_forward(C.byref(batch.cfg), _ptr(self._ktheta()), _ptr(batch.pos),
                                                 _ptr(batch.charges), _ptr(batch.bags), _ptr(batch.actions),
                                                 _ptr(self.leb), _ptr(ws), ws.numel(), _ptr(out), self._s()))
        self._last_ws, self._last_cfg = ws, batch.cfg
        # every (cfg, workspace) a training forward used since the last check: ppo.train keeps three mini-batches in flight on
        # their own workspaces, and the list build raises its error flags in the workspace of the mini-batch it belongs to
        self.__dict__.setdefault('_unchecked', {})[slot] = (batch.cfg, ws)
        return out

    def check_inputs(self) -> None:
        """Surface the list-build flags of EVERY training forward since the last call (mg_cov_check: canvases whose atoms are
        not compacted to the front, TA / TE that do not match the charges) as RuntimeError -- one (cfg, workspace) pair per
        workspace slot, i.e. the most recent mini-batch of each stream ppo.train keeps in flight.  It reads 16 bytes back per
        slot, i.e. synchronises: ppo.train calls it once per call, right after the first epoch's own device -> host copy.
        A rank that evaluated nothing since the last call checks nothing (no stale cfg against a workspace the rollout's
        sampling launches have since reused)."""
        pending = self.__dict__.get('_unchecked') or {}
        self._unchecked = {}
        with self._guard():
            for cfg, ws in pending.values():
                self._chk(self._L().mg_cov_check(C.byref(cfg), _ptr(ws), ws.numel(), self._s()))

    def input_flags_async(self) -> Optional[torch.Tensor]:
        """the list-build flags check_inputs() would read, as ONE pinned host tensor filled by asynchronous copies on the current
        stream (one int per workspace slot used since the last check; None if there is nothing to check): the caller waits for
        an event it records behind this call and hands the tensor to `raise_input_flags` -- ppo.train's run-ahead loop never
        drains the stream for the check."""
        pending = list((self.__dict__.get('_unchecked') or {}).values())
        self._unchecked = {}
        if not pending:
            return None
        host = torch.empty(len(pending), dtype=torch.int32).pin_memory()
        for k, (cfg, ws) in enumerate(pending):
            off, cnt = C.c_int64(), C.c_int64()
            self._chk(self._L().mg_cov_workspace_lookup(C.byref(cfg), b'err', C.byref(off), C.byref(cnt)))
            host[k:k + 1].copy_(ws.view(torch.int32)[off.value:off.value + 1], non_blocking=True)
        return host

    @staticmethod
    def raise_input_flags(flags: Optional[torch.Tensor]) -> None:
        for flag in ([] if flags is None else flags.tolist()):
            if flag == 1:
                raise RuntimeError('molgym_hip error -1: charges: the real atoms of a canvas must be compacted to the front '
                                   '(a padding slot precedes an atom)')
            if flag == 2:
                raise RuntimeError('molgym_hip error -1: cfg.TA / cfg.TE do not match the atom counts in charges')
            if flag != 0:
                raise RuntimeError(f'molgym_hip error -1: list build reported error {flag}')

    def ppo_minibatch(self, batch: 'DeviceBatch', clip_ratio: float, vf_coef: float, entropy_coef: float,
                      loss_scale: float = 1.0, slot: int = 0, stats_accum: Optional[torch.Tensor] = None,
                      graph: Optional[bool] = None, epoch_cache: bool = False) -> torch.Tensor:
        """One compute_loss forward + backward (molgym/ppo.py:124-131) entirely on the device, ONE C call (mg_cov_ppo_step):
        step -> float64 PPO loss -> hand-written backward, gradients ACCUMULATED (atomically) into theta.grad.
        Returns the 6 float64 loss statistics (device tensor, no sync); `stats_accum` (6 float64, optional) additionally gets
        loss_scale x statistics added on the device (ppo.train's epoch mean without a tensor per mini-batch).  `slot` selects an
        independent workspace AND an independent cached graph, so the mini-batches of one epoch -- independent given theta --
        can be in flight on several HIP streams.  `graph` (default: on, MG_GRAPH=0 in the environment turns it off): the ~27
        launches of the step are recorded and issued as one hipGraph launch whose kernel nodes are updated in place
        (include/molgym_hip.h: the update loop is otherwise bound by the host's launch rate); the library falls back to plain
        stream launches where a graph cannot express the step.  `epoch_cache` (ppo.train's loop): theta is constant over the
        mini-batches of an epoch, so the derived weight matrices of this slot's workspace are prepared by the slot's FIRST
        mini-batch after `invalidate_weights()` only, and the expanded complex weight gradients stay in the workspace until
        `fold_gradients()` -- theta.grad is incomplete until then."""
        lib = self._L()
        ws = self._workspace(batch.cfg, slot, epoch_step=epoch_cache)
        flags = 0
        if epoch_cache:
            st = self.__dict__.setdefault('_ws_epoch', {}).setdefault(slot, {'weights': False, 'pending': False})
            flags = _lib.STEP_DEFER_FOLD | (_lib.STEP_WEIGHTS_CURRENT if st['weights'] else 0)
            st.update(weights=True, pending=True, cfg=batch.cfg)
        else:
            self.__dict__.get('_ws_epoch', {}).pop(slot, None)  # (this call re-prepares the weights and zeroes the accumulator)
        B = batch.cfg.B
        dev = self.theta.device
        out = torch.empty(3, B, dtype=torch.float32, device=dev)
        gout = torch.empty(3, B, dtype=torch.float32, device=dev)
        stats = torch.empty(6, dtype=torch.float64, device=dev)
        if self.theta.grad is None:
            self.theta.grad = torch.zeros_like(self.theta)
        use_graph = self.use_graphs if graph is None else graph
        used = C.c_int32(0)
        theta_k = self._ktheta(refresh=not (flags & _lib.STEP_WEIGHTS_CURRENT))
        with self._guard():
            self._chk(lib.mg_cov_ppo_step(C.byref(batch.cfg), _ptr(theta_k), _ptr(batch.pos), _ptr(batch.charges),
                                          _ptr(batch.bags), _ptr(batch.actions), _ptr(self.leb), _ptr(ws), ws.numel(),
                                          _ptr(batch.logp), _ptr(batch.adv), _ptr(batch.ret), clip_ratio, vf_coef, entropy_coef,
                                          float(loss_scale), _ptr(out), _ptr(gout), _ptr(stats), _ptr(stats_accum),
                                          _ptr(self._kgrad(slot)), slot if use_graph else -1, flags, C.byref(used), self._s()))
        if not epoch_cache:
            self._kgrad_collect(slot)  # (maxl < 4; with epoch_cache the gather follows the fold, fold_gradients)
        self._last_ws, self._last_cfg, self._last_out = ws, batch.cfg, out
        self.last_step_used_graph = bool(used.value)
        self.__dict__.setdefault('_unchecked', {})[slot] = (batch.cfg, ws)
        return stats

    def _step_sample(self, observations: List[ObservationType]) -> Dict[str, Any]:
        """step(obs) of the rollout (agent.py:229-292, ppo.py:188,205): draw the sub-actions on the device
        (training) or take the argmax variants (evaluation) and return logp / ent / v of what was drawn."""
        N = self.observation_space.canvas_space.size
        pos, charges, bags, natoms = parse_observations_host(observations, self.zs, N)
        B = len(observations)
        cfg = self._make_cfg(B, natoms)
        dev = self.theta.device
        d_pos, d_chg, d_bag = self._upload_packed(pos, charges, bags)
        ws = self._workspace(cfg)
        out = torch.empty(3, B, dtype=torch.float32, device=dev)
        acts = torch.empty(B, 6, dtype=torch.float32, device=dev)
        seed = int(torch.randint(0, 2**62, (1, )).item())  # follows torch.manual_seed (util.set_seeds)
        mode = 1 if self.training else 2
        with self._guard():
            self._chk(self._L().mg_cov_sample(C.byref(cfg), _ptr(self._ktheta()), _ptr(d_pos), _ptr(d_chg), _ptr(d_bag),
                                                _ptr(self.leb), C.c_uint64(seed), mode, _ptr(ws), ws.numel(),
                                                _ptr(acts), _ptr(out), self._s()))
        self._last_ws, self._last_cfg = ws, None
        self.__dict__.get('_unchecked', {}).pop(0, None)  # slot 0's workspace now holds this sampling launch's lists
        dists = self._dists(cfg, ws, d_bag)
        with self._guard():  # this path synchronises for the actions anyway: surface the list build's error flags
            self._chk(self._L().mg_cov_check(C.byref(cfg), _ptr(ws), ws.numel(), self._s()))
        host = acts.cpu().numpy()
        return {'actions': [self.to_action_space(a, o) for a, o in zip(host, observations)], 'a': acts,
                'logp': out[0], 'ent': out[1], 'v': out[2], 'dists': dists}

    @staticmethod
    def draw_seed() -> int:
        """one sampling seed from the torch RNG (follows torch.manual_seed, util.set_seeds)"""
        return int(torch.randint(0, 2**62, (1, )).item())

    # -- persistent device canvases (rollouts): no per-step parse, the drawn atom is appended on the device -------------
    def make_canvas(self, observations: List[ObservationType]) -> 'DeviceCanvas':
        from .canvas import DeviceCanvas
        return DeviceCanvas(self, observations)

    def step_canvas(self, canvas: 'DeviceCanvas', commit: bool = True, seed: Optional[int] = None,
                    sample_ids: Tuple[int, int] = (0, 1)) -> Dict[str, Any]:
        """step(observations) of the rollout on resident canvases: same sampling kernels, same return dict; with
        `commit` the drawn atoms are then placed on the canvases in HBM (the host hands `actions` to the environments
        and calls `canvas.sync` for the ones that were reset or changed in any other way).
        `seed` / `sample_ids = (base, stride)`: the random stream of canvas row b is keyed by (seed, base + stride * b) --
        a rollout stepped in groups of environments passes the step's common seed and the group's environment ids, and
        every environment draws what one call over all of them would draw (ppo._rollout_pipelined)."""
        B = canvas.E
        natoms_before = canvas.natoms.copy()
        cfg = self._make_cfg(B, natoms_before)
        dev = self.theta.device
        ws = self._workspace(cfg)
        out = torch.empty(3, B, dtype=torch.float32, device=dev)
        acts = torch.empty(B, 6, dtype=torch.float32, device=dev)
        if seed is None:
            seed = self.draw_seed()
        mode = 1 if self.training else 2
        with self._guard():
            self._chk(self._L().mg_cov_sample_ids(C.byref(cfg), _ptr(self._ktheta()), _ptr(canvas.pos32), _ptr(canvas.charges),
                                                    _ptr(canvas.bags), _ptr(self.leb), C.c_uint64(seed), int(sample_ids[0]),
                                                    int(sample_ids[1]), mode, _ptr(ws), ws.numel(), _ptr(acts), _ptr(out),
                                                    self._s()))
        self._last_ws, self._last_cfg = ws, None
        self.__dict__.get('_unchecked', {}).pop(0, None)
        dists = self._dists(cfg, ws, canvas.bags.clone())
        newpos = canvas.append(acts, commit)
        host_a, host_p = acts.cpu().numpy(), newpos.cpu().numpy()  # the action rows and the positions they place
        elements = np.rint(host_a[:, 1]).astype(np.int64)
        if commit:
            placed = (np.asarray(self.zs)[elements] != 0) & (natoms_before < cfg.N)
            canvas.natoms = natoms_before + placed.astype(np.int32)
            canvas.bags_host[np.nonzero(placed)[0], elements[placed]] -= 1
            canvas.last_placed = (placed.copy(), host_p.astype(np.float64))
        actions = [(int(e), (float(p[0]), float(p[1]), float(p[2]))) for e, p in zip(elements, host_p)]
        return {'actions': actions, 'a': acts, 'logp': out[0], 'ent': out[1], 'v': out[2], 'dists': dists}

    def workspace_view(self, name: str, cfg: _lib.CovCfg) -> torch.Tensor:
        """float32 view of a named intermediate of the last forward (tests only)."""
        off, cnt = C.c_int64(), C.c_int64()
        self._chk(self._L().mg_cov_workspace_lookup(C.byref(cfg), name.encode(), C.byref(off), C.byref(cnt)))
        return self._last_ws.view(torch.float32)[off.value:off.value + cnt.value]

    def workspace_view_int(self, name: str, cfg: _lib.CovCfg) -> torch.Tensor:
        """int32 view of a named index list of the last forward (natoms, atom_off, err, ...)."""
        return self.workspace_view(name, cfg).view(torch.int32)
