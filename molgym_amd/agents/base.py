"""Actor-critic interface (mirror of /root/reference/molgym/agents/base.py:10-19) and what the two HIP agents share:
ONE flat float32 parameter vector `theta` whose named slots are the reference module's state_dict entries."""
import abc
import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

try:
    from torch.nn.modules.module import _IncompatibleKeys
except ImportError:  # pragma: no cover
    from collections import namedtuple
    _IncompatibleKeys = namedtuple('IncompatibleKeys', ['missing_keys', 'unexpected_keys'])


class AbstractActorCritic(torch.nn.Module, abc.ABC):
    def __init__(self, observation_space, action_space):
        super().__init__()
        self.observation_space = observation_space
        self.action_space = action_space

    @abc.abstractmethod
    def step(self, observations: List, actions: Optional[np.ndarray] = None) -> dict:
        raise NotImplementedError

    def evaluate_actions(self, observations: List, actions: np.ndarray) -> dict:
        """Action evaluation under the name BASELINE.json's north_star uses; the reference spells it
        step(observations, actions) (base.py:17-19, ppo.py:26)."""
        return self.step(observations, actions)


class FlatThetaAgent(AbstractActorCritic):
    """`self.theta` (nn.Parameter, flat) + `self.slot_table` (name -> (offset, shape), molgym_amd/layout.py).

    state_dict() / load_state_dict() speak the REFERENCE module's keys -- one entry per tensor the reference agent
    registers (covariant/agent.py:58-143, internal/agent.py:37-105) -- so a checkpoint written through state_dict()
    by either implementation loads into the other; the flat vector itself is accepted too ({'theta': ...}).
    Whole-module pickling (ModelIO, tools/model_util.py:82-100) keeps working: only caches are dropped."""

    # entries of a reference checkpoint that carry no trainable state of this path (cormorant's fixed cut-off
    # parameters and zero buffers, the agents' index helper buffers)
    _IGNORED_SUFFIXES = ('soft_cut_rad', 'soft_cut_width', 'zero', 'channel_offsets', 'zs_tensor', 'leb')

    def _slot(self, name: str, tensor: Optional[torch.Tensor] = None) -> torch.Tensor:
        off, shape = self.slot_table[name]
        t = self.theta if tensor is None else tensor
        return t[off:off + int(np.prod(shape))].view(shape)

    def export_state_dict(self) -> Dict[str, torch.Tensor]:
        """Named CLONES with the reference module's state_dict keys."""
        t = self.theta.detach()
        return {k: self._slot(k, t).clone() for k in self.slot_table}

    def import_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        with torch.no_grad():
            for k in self.slot_table:
                self._slot(k).copy_(sd[k].to(self.theta).reshape(self.slot_table[k][1]))

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        if args:  # the legacy positional form (destination, prefix, keep_vars) torch.nn.Module still accepts
            destination = args[0] if destination is None else destination
            prefix = args[1] if len(args) > 1 and prefix == '' else prefix
            keep_vars = args[2] if len(args) > 2 and keep_vars is False else keep_vars
        out = OrderedDict() if destination is None else destination
        t = self.theta if keep_vars else self.theta.detach()
        for k in self.slot_table:
            out[prefix + k] = self._slot(k, t)
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = dict(state_dict)
        if set(sd) == {'theta'}:
            with torch.no_grad():
                self.theta.copy_(sd['theta'].to(self.theta).reshape(-1))
            return _IncompatibleKeys([], [])
        missing = [k for k in self.slot_table if k not in sd]
        unexpected = [k for k in sd if k not in self.slot_table and not k.endswith(self._IGNORED_SUFFIXES)]
        bad = [k for k, (_, s) in self.slot_table.items() if k in sd and tuple(sd[k].shape) != tuple(s)]
        if bad:
            raise RuntimeError('size mismatch for ' + ', '.join(
                f'{k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(self.slot_table[k][1])}' for k in bad))
        if strict and (missing or unexpected):
            raise RuntimeError(f'Error(s) in loading state_dict for {type(self).__name__}: missing keys {missing}, '
                               f'unexpected keys {unexpected}')
        with torch.no_grad():
            for k in self.slot_table:
                if k in sd:
                    self._slot(k).copy_(sd[k].to(self.theta))
        return _IncompatibleKeys(missing, unexpected)

    # -- device plumbing ----------------------------------------------------------------------------------------------
    def _upload_packed(self, *arrays: np.ndarray) -> List[torch.Tensor]:
        """ONE host -> device copy for several 4-byte-typed (float32 / int32) arrays: packed into one host buffer, sections
        aligned to 256 bytes, returned as typed device views of the arrays' shapes (one H2D per batch; four
        separate pageable copies cost four synchronous transfers)."""
        offs, total = [], 0
        for x in arrays:
            assert x.dtype.itemsize == 4, x.dtype
            offs.append(total)
            total += (x.size + 63) // 64 * 64
        host = np.empty(max(total, 64), dtype=np.int32)
        for x, o in zip(arrays, offs):
            host[o:o + x.size] = np.ascontiguousarray(x).reshape(-1).view(np.int32)
        dev = torch.from_numpy(host).to(self.theta.device)
        out = []
        for x, o in zip(arrays, offs):
            v = dev[o:o + x.size]
            out.append((v.view(torch.float32) if x.dtype == np.float32 else v).view(x.shape))
        return out

    def _guard(self):
        """Every C call runs with the agent's device current: the library keeps per-device state (CG tables in
        __constant__ memory, function attributes, side stream) keyed by hipGetDevice()."""
        return torch.cuda.device(self.theta.device)

    def _s(self):
        return C.c_void_p(torch.cuda.current_stream(self.theta.device).cuda_stream)

    def adam_supported(self, optimizer) -> bool:
        """whether `adam_step` takes this optimizer (a plain torch.optim.Adam over the single flat theta, no step hooks)"""
        if type(optimizer) is not torch.optim.Adam or len(optimizer.param_groups) != 1:
            return False
        # step hooks registered on the optimizer (or globally) must run: leave those cases to optimizer.step()
        from torch.optim import optimizer as _opt_mod
        if getattr(optimizer, '_optimizer_step_pre_hooks', None) or getattr(optimizer, '_optimizer_step_post_hooks', None) or \
                getattr(_opt_mod, '_global_optimizer_pre_hooks', None) or getattr(_opt_mod, '_global_optimizer_post_hooks', None):
            return False
        g = optimizer.param_groups[0]
        p = self.theta
        if len(g['params']) != 1 or g['params'][0] is not p or p.grad is None or not p.is_cuda:
            return False
        if g.get('capturable') or g.get('fused') or g.get('differentiable') or torch.is_tensor(g['lr']):
            return False
        if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse or not p.grad.is_contiguous():
            return False
        if g['amsgrad'] and len(optimizer.state[p]) and 'max_exp_avg_sq' not in optimizer.state[p]:
            return False
        return True

    def adam_step(self, optimizer, skip_flag: Optional[torch.Tensor] = None) -> bool:
        """optimizer.step() (ppo.py:145) for a plain torch.optim.Adam over the flat theta as ONE HIP launch (mg_adam_step)
        on the optimizer's own state tensors, which are created exactly as torch creates them -- so state_dict(), learning-
        rate schedulers (their step counter is advanced here) and a later optimizer.step() see nothing unusual; an
        optimizer with step hooks is left to optimizer.step().  Returns False (nothing done) for anything else:
        another optimizer class, several parameter groups / tensors, fused / capturable / differentiable variants.
        `skip_flag` (device int32, optional): the launch does nothing when it is non-zero (mg_adam_step_gated: the KL test of
        ppo.train taken on the device); the optimizer's step counter is advanced either way -- the caller takes the skipped
        steps back (`adam_unstep`) once it has read the flag."""
        from .. import _lib
        if not self.adam_supported(optimizer):
            return False
        g = optimizer.param_groups[0]
        p = self.theta
        st = optimizer.state[p]
        if len(st) == 0:  # torch.optim.Adam._init_group
            st['step'] = torch.tensor(0.0, dtype=torch.float64 if torch.get_default_dtype() == torch.float64 else torch.float32)
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if g['amsgrad']:
                st['max_exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if g['amsgrad'] and 'max_exp_avg_sq' not in st:
            return False
        st['step'] += 1
        step = int(st['step'].item()) if torch.is_tensor(st['step']) else int(st['step'])
        beta1, beta2 = g['betas']
        ptr = lambda t: C.c_void_p(t.data_ptr())
        with self._guard():
            _lib.check(_lib.lib().mg_adam_step_gated(p.numel(), ptr(p.data), ptr(p.grad), ptr(st['exp_avg']), ptr(st['exp_avg_sq']),
                                                     ptr(st['max_exp_avg_sq']) if g['amsgrad'] else None, float(g['lr']),
                                                     float(beta1), float(beta2), float(g['eps']), float(g['weight_decay']), step,
                                                     1 if g.get('maximize') else 0,
                                                     None if skip_flag is None else ptr(skip_flag), self._s()))
        # what a learning-rate scheduler's wrapper of optimizer.step() would have recorded (lr_scheduler.py: `_step_count`,
        # `_opt_called`; the scheduler warns about "lr_scheduler.step() before optimizer.step()" otherwise)
        if hasattr(optimizer, '_step_count'):
            optimizer._step_count += 1
        optimizer._opt_called = True
        return True

    def adam_unstep(self, optimizer, count: int) -> None:
        """take back `count` optimizer steps whose launches the device-side gate skipped (step counters only: the parameters
        and moments were never touched)"""
        if count <= 0:
            return
        st = optimizer.state[self.theta]
        st['step'] -= count
        if hasattr(optimizer, '_step_count'):
            optimizer._step_count -= count

    def ppo_epoch_end(self, max_norm: float, stats_accum: torch.Tensor, num_minibatches: int, kl_limit: float,
                      rec: torch.Tensor, stop_flag: torch.Tensor) -> None:
        """mg_ppo_epoch_end: KL test, gradient norm and clip of one PPO epoch on the device (ppo.py:133-144); `rec` (8 float64)
        receives the epoch's statistics, the pre-clip norm and the state of the latching `stop_flag` (int32)."""
        from .. import _lib
        scratch = torch.empty(1, dtype=torch.float32, device=self.theta.device)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        with self._guard():
            _lib.check(_lib.lib().mg_ppo_epoch_end(self.theta.numel(), ptr(self.theta.grad), float(max_norm), ptr(stats_accum),
                                                   1.0 / max(int(num_minibatches), 1), float(kl_limit), ptr(rec), ptr(stop_flag),
                                                   ptr(scratch), self._s()))

    def grad_norm_clip(self, max_norm: float = 0.0) -> torch.Tensor:
        """||theta.grad||_2 as a 1-element device tensor (util.compute_gradient_norm, tools/util.py:61-69) and, if
        max_norm > 0, theta.grad *= min(1, max_norm / (norm + 1e-6)) (clip_grad_norm_, ppo.py:144) -- one C call on
        the flat gradient instead of ~60 per-tensor kernels."""
        from .. import _lib
        norm = torch.empty(2, dtype=torch.float32, device=self.theta.device)
        with self._guard():
            _lib.check(_lib.lib().mg_grad_norm_clip(self.theta.numel(), C.c_void_p(self.theta.grad.data_ptr()),
                                                    float(max_norm), C.c_void_p(norm.data_ptr()), self._s()))
        return norm[:1]
